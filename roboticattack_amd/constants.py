"""Constants of the UADA/UPA/TMA hot path (all from the reference; file:line cited per item)."""
from __future__ import annotations

IMG = 224  # camera frame side; reference hard-codes 224x224 inputs (UADA.py:60, modeling_prismatic.py:120)
IGNORE_INDEX = -100  # UADA.py:20
CANVAS_SENTINEL = -100.0  # appply_random_transform.py:111
KEEP_THRESHOLD = -20.0  # appply_random_transform.py:131  (canvas < -20 -> camera pixel, else patch pixel)

# Dual normalisation statistics: DINOv2 then SigLIP (UADA.py:56-57).
MEAN0 = (0.484375, 0.455078125, 0.40625)
STD0 = (0.228515625, 0.2236328125, 0.224609375)
MEAN1 = (0.5, 0.5, 0.5)
STD1 = (0.5, 0.5, 0.5)
MEAN6 = MEAN0 + MEAN1
STD6 = STD0 + STD1

# Token layout (prismatic/vla/action_tokenizer.py:31-36, configuration_prismatic.py:86).
TOKENIZER_VOCAB = 32000
MODEL_VOCAB = 32064  # Llama vocab padded to a multiple of 64
N_BINS = 256
ACTION_TOKEN_BEGIN_IDX = TOKENIZER_VOCAB - (N_BINS + 1)  # 31743
ACTION_LO = 31744  # first action token (bin 256, action +0.996)
ACTION_HI = 32000  # one past the last action token
ACTION_MID = 31872  # UADA.py:391 split point of the "opposite extreme" target
EOS_ID = 2
BOS_ID = 1
PAD_ID = 32000  # UADA.py:43
N_IMG_TOKENS = 256  # ViT patch tokens inserted after BOS (modeling_prismatic.py:383-385)

# Random geometry bounds (appply_random_transform.py:11-13, :82).
MAX_ANGLE_DEG = 30.0
MAX_SHEAR = 0.2
P_IDENTITY = 0.2
