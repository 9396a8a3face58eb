"""Import harness for the READ-ONLY upstream reference (this container only).

Used exclusively by tools/gen_golden.py to run the reference's own Python
functions on CPU and record golden input/output vectors under tests/golden/.
Nothing here (and nothing under /root/reference) travels to the GPU box, and
no product / test / bench code imports this module.

The reference as shipped does not import:
  * D1  appply_random_transform.py:43 has a 3-space `def` (IndentationError);
        repaired IN MEMORY by prepending one space to that line.
  * torchvision / wandb / seaborn / timm are absent from this image -> stubbed
    with the minimal behaviour the hot path uses (ToTensor = u8/255 CHW f32,
    ToPILImage = mul(255).byte()).
  * transformers 5.x dropped `AdamW` and `AutoModelForVision2Seq` -> the caller
    installs a restated HF-4.40.1 AdamW (third-party, "parity unpinned").
"""
from __future__ import annotations

import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("VAA_REFERENCE_ROOT", "/root/reference")


def _stub(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package so `import a.b` works
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load_by_path(modname: str, path: str) -> types.ModuleType:
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


class _ToTensor:
    """torchvision.transforms.ToTensor for a PIL RGB image: u8 HWC -> f32 CHW / 255."""

    def __call__(self, pic):
        arr = np.asarray(pic, dtype=np.uint8)
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
        return t.to(torch.float32).div(255)


class _ToPILImage:
    def __call__(self, t):
        from PIL import Image

        arr = t.detach().cpu().mul(255).byte().permute(1, 2, 0).numpy()
        return Image.fromarray(arr)


class _Resize:
    def __init__(self, size):
        self.size = size

    def __call__(self, t):
        import torch.nn.functional as F

        return F.interpolate(t[None], size=self.size, mode="bilinear", antialias=True, align_corners=False)[0]


def load_reference():
    """Returns a namespace with the reference modules: transform, UADA, UADA_ddp, UPA, TMA, action_tokenizer."""
    import transformers  # noqa: F401  (must be imported before the stubs: it probes find_spec)
    import transformers.modeling_outputs  # noqa: F401

    tv_t = _stub("torchvision.transforms", ToTensor=_ToTensor, ToPILImage=_ToPILImage, Resize=_Resize)
    _stub("torchvision", transforms=tv_t)
    _stub("wandb", log=lambda *a, **k: None, Image=lambda *a, **k: None, init=lambda *a, **k: None)
    _stub("seaborn", set_theme=lambda *a, **k: None)
    for pkg in (
        "prismatic",
        "prismatic.vla",
        "prismatic.models",
        "prismatic.models.backbones",
        "prismatic.models.backbones.llm",
        "prismatic.util",
        "prismatic.extern",
        "prismatic.extern.hf",
        "white_patch",
    ):
        _stub(pkg)
    _stub("prismatic.extern.hf.configuration_prismatic", OpenVLAConfig=object)
    _stub("prismatic.extern.hf.processing_prismatic", PrismaticProcessor=object)
    _stub("prismatic.extern.hf.modeling_prismatic", OpenVLAForActionPrediction=object)
    _stub("white_patch.openvla_dataloader", get_dataset=None, get_dataloader=None)

    at = _load_by_path("prismatic.vla.action_tokenizer", f"{REF}/prismatic/vla/action_tokenizer.py")
    _load_by_path(
        "prismatic.models.backbones.llm.prompting",
        f"{REF}/prismatic/models/backbones/llm/prompting/base_prompter.py",
    )
    _load_by_path("prismatic.util.data_utils", f"{REF}/prismatic/util/data_utils.py")

    # D1: in-memory one-character indentation repair of line 43, nothing else.
    src = open(f"{REF}/VLAAttacker/white_patch/appply_random_transform.py").read().split("\n")
    assert src[42].startswith("   def simulation_random_patch"), src[42]
    src[42] = " " + src[42]
    tr = types.ModuleType("appply_random_transform")
    tr.__spec__ = importlib.machinery.ModuleSpec("appply_random_transform", None)
    exec(compile("\n".join(src), f"{REF}/VLAAttacker/white_patch/appply_random_transform.py", "exec"), tr.__dict__)
    sys.modules["appply_random_transform"] = tr
    sys.modules["white_patch.appply_random_transform"] = tr

    import transformers as _tf

    if not hasattr(_tf, "AutoModelForVision2Seq"):
        _tf.AutoModelForVision2Seq = object

    wp = f"{REF}/VLAAttacker/white_patch"
    ns = types.SimpleNamespace(
        transform=tr,
        action_tokenizer=at,
        UADA=_load_by_path("white_patch.UADA", f"{wp}/UADA.py"),
        UPA=_load_by_path("white_patch.UPA", f"{wp}/UPA.py"),
        TMA=_load_by_path("white_patch.TMA", f"{wp}/TMA.py"),
        UADA_ddp=_load_by_path("white_patch.UADA_ddp", f"{wp}/UADA_ddp.py"),
    )
    return ns


def load_transform_d2_repaired():
    """A second, independent module object of the reference transform with D1 + D2 repaired in memory (see the header).
    `load_reference()` must have been called first (it installs the torchvision stubs)."""
    path = f"{REF}/VLAAttacker/white_patch/appply_random_transform.py"
    src = open(path).read().split("\n")
    assert src[42].startswith("   def simulation_random_patch"), src[42]
    src[42] = " " + src[42]
    # D2: lines 104-118 (1-based). Assert the shipped text before touching it so a changed reference is noticed.
    assert src[104].strip() == "modified_images = []", src[104]
    assert src[106].strip() == "for im in images:", src[106]
    assert "int(patch_height * scale), int(patch_width * scale)" in src[114], src[114]
    assert src[115].strip() == "patch = transforms.Resize((height, width))(patch)", src[115]
    src[104] = src[104] + "; base_patch = patch"
    src[114] = src[114].replace("int(patch_height * scale), int(patch_width * scale)",
                                "int(base_patch.shape[1] * scale), int(base_patch.shape[2] * scale)")
    src[115] = src[115].replace("(patch)", "(base_patch)")
    tr = types.ModuleType("appply_random_transform_d2")
    tr.__spec__ = importlib.machinery.ModuleSpec("appply_random_transform_d2", None)
    exec(compile("\n".join(src), path, "exec"), tr.__dict__)
    return tr


class FakeTokenizer:
    """Stands in for the Llama tokenizer: only vocab_size is used by ActionTokenizer's numeric paths."""

    vocab_size = 32000
    pad_token_id = 32000
    model_max_length = 2048
