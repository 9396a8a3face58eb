"""Seeded synthetic BridgeData-shaped inputs.

The reference's data layer (TF/dlimp RLDS, openvla_dataloader.py:81) is out of scope; what the
hot loop consumes is the batch dict produced by `PaddedCollatorForActionPrediction.__call__`
(prismatic/util/data_utils.py:101-145) on top of `RLDSBatchTransform.__call__`
(prismatic/vla/datasets/datasets.py:40-69):

    pixel_values   : list of B PIL RGB 224x224 images (u8)
    input_ids      : int64 [B, L]  = [BOS, prompt tokens..., a1..a7, EOS], right-padded with 32000
    labels         : int64 [B, L]  = -100 except the last 8 real tokens (7 action tokens + EOS)
    attention_mask : bool  [B, L]  = input_ids != 32000

All generators use numpy's legacy RandomState (bit-stable across platforms) so that the golden
fixtures made in the survey container can be re-derived on the GPU box from the seed alone.
"""
from __future__ import annotations

import numpy as np
import torch

from .constants import ACTION_HI, ACTION_LO, BOS_ID, EOS_ID, IGNORE_INDEX, IMG, PAD_ID


def synth_images(seed: int, batch: int, kind: str = "noise") -> np.ndarray:
    """u8 [B,224,224,3] HWC frames. `noise`: iid uniform bytes; `smooth`: low-frequency ramps + mild noise."""
    rs = np.random.RandomState(seed)
    if kind == "noise":
        return rs.randint(0, 256, (batch, IMG, IMG, 3), dtype=np.uint8)
    if kind == "smooth":
        yy, xx = np.meshgrid(np.arange(IMG), np.arange(IMG), indexing="ij")
        out = np.empty((batch, IMG, IMG, 3), dtype=np.uint8)
        for b in range(batch):
            ph = rs.uniform(0, 2 * np.pi, size=3)
            fx = rs.uniform(0.01, 0.05, size=3)
            fy = rs.uniform(0.01, 0.05, size=3)
            for c in range(3):
                v = 127.5 + 100.0 * np.sin(fx[c] * xx + fy[c] * yy + ph[c])
                v = v + rs.randint(-8, 9, size=(IMG, IMG))
                out[b, :, :, c] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
        return out
    raise ValueError(f"unknown image kind {kind!r}")


def to_pil_list(images_u8: np.ndarray):
    """What the reference collator hands to the attack loop: a python list of PIL RGB images."""
    from PIL import Image

    return [Image.fromarray(np.ascontiguousarray(im)) for im in images_u8]


def synth_text_batch(seed: int, batch: int, min_len: int = 28, max_len: int = 44):
    """(input_ids, labels, attention_mask) shaped like the reference collator's output."""
    rs = np.random.RandomState(seed)
    lens = rs.randint(min_len, max_len + 1, size=batch)
    L = int(lens.max())
    input_ids = np.full((batch, L), PAD_ID, dtype=np.int64)
    labels = np.full((batch, L), IGNORE_INDEX, dtype=np.int64)
    for b in range(batch):
        n = int(lens[b])
        prompt = rs.randint(3, ACTION_LO, size=n - 9)
        actions = rs.randint(ACTION_LO, ACTION_HI, size=7)
        row = np.concatenate([[BOS_ID], prompt, actions, [EOS_ID]])
        input_ids[b, :n] = row
        labels[b, n - 8 : n] = row[n - 8 :]
    input_ids_t = torch.from_numpy(input_ids)
    return input_ids_t, torch.from_numpy(labels), input_ids_t.ne(PAD_ID)


def synth_batch(seed: int, batch: int, kind: str = "noise", as_pil: bool = True, **kw) -> dict:
    """A full synthetic batch dict with the reference collator's keys."""
    imgs = synth_images(seed, batch, kind)
    input_ids, labels, attention_mask = synth_text_batch(seed + 7919, batch, **kw)
    return {
        "pixel_values": to_pil_list(imgs) if as_pil else imgs,
        "input_ids": input_ids,
        "labels": labels,
        "attention_mask": attention_mask,
        "instructions": ["synthetic"] * batch,
        "dataset_names": ["synthetic"] * batch,
    }


class SyntheticLoader:
    """Infinite iterable of synthetic batches; stands in for the RLDS DataLoader (openvla_dataloader.py:81)."""

    def __init__(self, batch: int, seed: int = 1234, kind: str = "noise", as_pil: bool = True, length: int | None = None):
        self.batch, self.seed, self.kind, self.as_pil, self.length = batch, seed, kind, as_pil, length

    def __len__(self):
        return self.length if self.length is not None else 1 << 30

    def __iter__(self):
        i = 0
        while self.length is None or i < self.length:
            yield synth_batch(self.seed + 104729 * i, self.batch, self.kind, self.as_pil)
            i += 1


def synth_upstream_grad(seed: int, batch: int, scale: float = 1e-3) -> torch.Tensor:
    """Seeded bf16 [B,6,224,224] stand-in for dL/d(pixel_values) coming back from the frozen model."""
    rs = np.random.RandomState(seed)
    g = rs.standard_normal((batch, 6, IMG, IMG)).astype(np.float32) * np.float32(scale)
    return torch.from_numpy(g).to(torch.bfloat16)


def synth_logits(seed: int, batch: int, seq: int, vocab: int, scale: float = 2.0, action_boost: float = 3.0) -> torch.Tensor:
    """Seeded f32 [B,S,V] logits; the 256 action columns get extra spread so the soft-argmax is non-trivial."""
    rs = np.random.RandomState(seed)
    z = rs.standard_normal((batch, seq, vocab)).astype(np.float32) * np.float32(scale)
    z[:, :, ACTION_LO:ACTION_HI] += rs.standard_normal((batch, seq, ACTION_HI - ACTION_LO)).astype(np.float32) * np.float32(
        action_boost
    )
    return torch.from_numpy(z)
