"""ctypes loader for libvaa_hip.so (C-ABI declared in include/vaa.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAA_LIB_PATH") or os.path.join(_HERE, "libvaa_hip.so")  # override = A/B experiments only

# mirrors include/vaa.h
VAA_OK = 0
MASK_LT_M20, MASK_NE_M100 = 0, 1
LOSS_UADA, LOSS_UADA_DDP, LOSS_UPA, LOSS_CE = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16 = 0, 1
LAYOUT_FULL, LAYOUT_ROWS = 0, 1
GRAD_FULL, GRAD_SLICE = 0, 1
OPT_ADAMW_HF, OPT_PGD_SIGN = 0, 1

MODEL_OP_EXPORTS = ("vaa_model_rope", "vaa_model_swiglu_fwd", "vaa_model_swiglu_bwd", "vaa_model_rmsnorm_fwd", "vaa_model_rmsnorm_bwd",
                    "vaa_model_attention_fwd", "vaa_model_attention_bwd", "vaa_model_layernorm_fwd", "vaa_model_layernorm_bwd", "vaa_model_scale_add")

EXPORTS = (
    "vaa_last_error",
    "vaa_version",
    "vaa_device_check",
    "vaa_patch_apply_fwd",
    "vaa_patch_grad_ws_bytes",
    "vaa_patch_grad_gather",
    "vaa_loss_ws_bytes",
    "vaa_loss_fwd_bwd",
    "vaa_loss_fwd_bwd_ex",
    "vaa_patch_update",
    "vaa_patch_apply_eval",
    "vaa_patch_embed_grad_ws_bytes",
    "vaa_patch_embed_grad_gather",
    "vaa_patch_embed_packed_elems",
    "vaa_patch_embed_pack_weights",
    "vaa_patch_embed_grad_multi_ws_bytes",
    "vaa_patch_embed_grad_gather_multi",
    "vaa_patch_resize_fwd",
    "vaa_patch_resize_ws_bytes",
    "vaa_patch_resize_bwd",
    "vaa_patch_apply_fwd_multi",
    "vaa_patch_grad_gather_multi",
    "vaa_loss_rowmap_bytes",
    "vaa_loss_rowmap_build",
    "vaa_loss_rows_ws_bytes",
    "vaa_loss_rows_fwd_bwd",
    "vaa_patch_apply_fwd_tiles",
    "vaa_patch_grad_partials",
    "vaa_patch_embed_grad_gather_multi_tiles",
    "vaa_patch_embed_grad_gather_tiles",
    "vaa_loss_rows_stats",
    "vaa_head_loss_ws_bytes",
    "vaa_head_loss_rows_applies",
    "vaa_head_loss_rows_stats",
    "vaa_head_loss_rows_finish",
    "vaa_head_slice_applies",
    "vaa_head_slice_ws_bytes",
    "vaa_head_slice_pack",
    "vaa_head_slice_fwd_bwd",
    "vaa_step_epilogue",
    "vaa_step_epilogue_update",
    "vaa_async_error",
    "vaa_prof_start",
    "vaa_prof_stop",
    "vaa_prof_get",
)


class VaaError(RuntimeError):
    pass


_lib = None


def _up_to_date() -> bool:
    """libvaa_hip.so exists and is newer than every source it is built from."""
    if not os.path.exists(LIB_PATH):
        return False
    t = os.path.getmtime(LIB_PATH)
    src = os.path.join(_HERE, "csrc")
    deps = [os.path.join(src, f) for f in os.listdir(src)] + [os.path.join(_HERE, "..", "include", f) for f in ("vaa.h", "vaa_model_ops.h")]
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build(verbose: bool = False, force: bool = True) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU). force=False skips the compile when the
    library is newer than its sources (what N ranks racing for a missing library need: one compiles, the others find it done)."""
    import fcntl
    import subprocess

    script = os.path.join(_HERE, "csrc", "build.sh")
    # rank-safe under torchrun: one process compiles (exclusive file lock), build.sh renames the finished file into place
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():  # another rank built it while this one waited for the lock
                return LIB_PATH
            out = subprocess.run(["bash", script], capture_output=True, text=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if out.returncode != 0:
        raise VaaError(f"building libvaa_hip.so failed:\n{out.stdout}\n{out.stderr}")
    if verbose:
        print(out.stdout.strip())
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("VAA_LIB_PATH") and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        # compiling the HIP extension is not a fallback: the product still runs only through libvaa_hip.so. Under torchrun every rank gets
        # here at once: build() takes the lock and re-checks, so ONE rank compiles and the others return as soon as the library is there
        build(force=False)
    if not os.path.exists(LIB_PATH):
        raise VaaError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or roboticattack_amd/csrc/build.sh"
        )
    try:
        # PyTorch-ROCm ships its own libamdhip64; if this library pulled in the system copy first, torch would later initialise a second
        # HIP runtime in the same process and see no GPU. Importing torch first makes both resolve to the one runtime torch loaded.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - the C-ABI is usable without torch
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.vaa_last_error.restype = C.c_char_p
    L.vaa_last_error.argtypes = []
    L.vaa_version.restype = i32
    L.vaa_device_check.restype = i32
    L.vaa_patch_apply_fwd.restype = i32
    L.vaa_patch_apply_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, vp]
    L.vaa_patch_grad_ws_bytes.restype = sz
    L.vaa_patch_grad_ws_bytes.argtypes = [i32, i32, i32]
    L.vaa_patch_grad_gather.restype = i32
    L.vaa_patch_grad_gather.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, sz, vp]
    L.vaa_patch_embed_grad_ws_bytes.restype = sz
    L.vaa_patch_embed_grad_ws_bytes.argtypes = [i32, i32, i32]
    L.vaa_patch_embed_packed_elems.restype = sz
    L.vaa_patch_embed_packed_elems.argtypes = [i32]
    L.vaa_patch_embed_pack_weights.restype = i32
    L.vaa_patch_embed_pack_weights.argtypes = [vp, i32, vp, vp]
    L.vaa_patch_embed_grad_gather.restype = i32
    L.vaa_patch_embed_grad_gather.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), i32, vp, vp, sz, vp]
    L.vaa_patch_embed_grad_multi_ws_bytes.restype = sz
    L.vaa_patch_embed_grad_multi_ws_bytes.argtypes = [i32]
    L.vaa_patch_embed_grad_gather_multi.restype = i32
    L.vaa_patch_embed_grad_gather_multi.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), i32, vp, vp, sz, vp]
    L.vaa_patch_resize_fwd.restype = i32
    L.vaa_patch_resize_fwd.argtypes = [vp, i32, i32, vp, i32, vp, vp]
    L.vaa_patch_resize_ws_bytes.restype = sz
    L.vaa_patch_resize_ws_bytes.argtypes = [i32, i32, i32]
    L.vaa_patch_resize_bwd.restype = i32
    L.vaa_patch_resize_bwd.argtypes = [vp, i32, i32, vp, i32, vp, vp, sz, vp]
    L.vaa_patch_apply_fwd_multi.restype = i32
    L.vaa_patch_apply_fwd_multi.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, vp]
    L.vaa_patch_grad_gather_multi.restype = i32
    L.vaa_patch_grad_gather_multi.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp]
    L.vaa_loss_ws_bytes.restype = sz
    L.vaa_loss_ws_bytes.argtypes = [i32, i32]
    L.vaa_loss_fwd_bwd.restype = i32
    L.vaa_loss_fwd_bwd.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, vp, vp, sz, vp]
    L.vaa_loss_fwd_bwd_ex.restype = i32
    L.vaa_loss_fwd_bwd_ex.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, vp, vp, vp, sz, vp]
    L.vaa_loss_rowmap_bytes.restype = sz
    L.vaa_loss_rowmap_bytes.argtypes = [i32, i32]
    L.vaa_loss_rowmap_build.restype = i32
    L.vaa_loss_rowmap_build.argtypes = [vp, i32, i32, vp, sz, vp]
    L.vaa_loss_rows_ws_bytes.restype = sz
    L.vaa_loss_rows_ws_bytes.argtypes = [i32]
    L.vaa_loss_rows_fwd_bwd.restype = i32
    L.vaa_loss_rows_fwd_bwd.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, vp, vp, i32, vp, sz, vp]
    L.vaa_patch_update.restype = i32
    L.vaa_patch_update.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, i32, f32, f32, vp, vp]
    L.vaa_patch_apply_eval.restype = i32
    L.vaa_patch_apply_eval.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.vaa_patch_apply_fwd_tiles.restype = i32
    L.vaa_patch_apply_fwd_tiles.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, vp, vp, vp, vp]
    L.vaa_patch_embed_grad_gather_multi_tiles.restype = i32
    L.vaa_patch_embed_grad_gather_multi_tiles.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), i32, vp, vp, sz, vp]
    L.vaa_patch_grad_partials.restype = i32
    L.vaa_patch_grad_partials.argtypes = [i32]
    L.vaa_patch_embed_grad_gather_tiles.restype = i32
    L.vaa_patch_embed_grad_gather_tiles.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, C.POINTER(f32), i32, vp, vp, sz, vp]
    L.vaa_loss_rows_stats.restype = i32
    L.vaa_loss_rows_stats.argtypes = [vp, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, i32, vp, sz, vp]
    L.vaa_head_loss_ws_bytes.restype = sz
    L.vaa_head_loss_ws_bytes.argtypes = [i32, i32]
    L.vaa_head_loss_rows_applies.restype = i32
    L.vaa_head_loss_rows_applies.argtypes = [i32, i32, i32]
    L.vaa_head_loss_rows_stats.restype = i32
    L.vaa_head_loss_rows_stats.argtypes = [vp, vp, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, sz, vp, sz, vp, vp]
    L.vaa_head_loss_rows_finish.restype = i32
    L.vaa_head_loss_rows_finish.argtypes = [vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, sz, vp, sz, vp, vp, vp, vp, vp]
    L.vaa_head_slice_applies.restype = i32
    L.vaa_head_slice_applies.argtypes = [i32, i32, i32]
    L.vaa_head_slice_ws_bytes.restype = sz
    L.vaa_head_slice_ws_bytes.argtypes = [i32]
    L.vaa_head_slice_pack.restype = i32
    L.vaa_head_slice_pack.argtypes = [vp, i32, i32, vp, vp]
    L.vaa_head_slice_fwd_bwd.restype = i32
    L.vaa_head_slice_fwd_bwd.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, vp, vp, sz, vp, vp, vp, vp, sz, vp]
    L.vaa_step_epilogue.restype = i32
    L.vaa_step_epilogue.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, sz, vp, vp, vp, vp, vp]
    L.vaa_step_epilogue_update.restype = i32
    L.vaa_step_epilogue_update.argtypes = [vp, i32, i32, vp, i32, i32, i32, i32, i32, C.POINTER(f32), vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32,
                                           i32, vp, vp]
    L.vaa_async_error.restype = i32
    L.vaa_async_error.argtypes = []
    L.vaa_prof_start.restype = i32
    L.vaa_prof_start.argtypes = [i32]
    L.vaa_prof_stop.restype = i32
    L.vaa_prof_stop.argtypes = []
    L.vaa_prof_get.restype = i32
    L.vaa_prof_get.argtypes = [i32, C.POINTER(C.c_char_p), C.POINTER(f32)]
    # optional model-side operators (include/vaa_model_ops.h)
    lng = C.c_long
    L.vaa_model_rope.restype = i32
    L.vaa_model_rope.argtypes = [vp, lng, lng, lng, vp, vp, i32, i32, i32, i32, f32, vp, vp]
    L.vaa_model_swiglu_fwd.restype = i32
    L.vaa_model_swiglu_fwd.argtypes = [vp, vp, vp, lng, vp]
    L.vaa_model_swiglu_bwd.restype = i32
    L.vaa_model_swiglu_bwd.argtypes = [vp, vp, vp, vp, vp, lng, vp]
    L.vaa_model_scale_add.restype = i32
    L.vaa_model_scale_add.argtypes = [vp, vp, vp, vp, lng, i32, vp]
    L.vaa_model_rmsnorm_fwd.restype = i32
    L.vaa_model_rmsnorm_fwd.argtypes = [vp, vp, vp, vp, lng, i32, f32, vp]
    L.vaa_model_rmsnorm_bwd.restype = i32
    L.vaa_model_rmsnorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, lng, i32, vp]
    L.vaa_model_layernorm_fwd.restype = i32
    L.vaa_model_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, lng, i32, f32, vp]
    L.vaa_model_layernorm_bwd.restype = i32
    L.vaa_model_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, lng, i32, vp]
    i64p = C.POINTER(C.c_int64)
    L.vaa_model_attention_fwd.restype = i32
    L.vaa_model_attention_fwd.argtypes = [vp, i64p, vp, i64p, vp, i64p, vp, i64p, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    L.vaa_model_attention_bwd.restype = i32
    L.vaa_model_attention_bwd.argtypes = [vp, i64p, vp, i64p, vp, i64p, vp, i64p, vp, i64p, vp, vp, vp, i64p, vp, i64p, vp, i64p, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != VAA_OK:
        msg = lib().vaa_last_error().decode("utf-8", "replace")
        raise VaaError(f"{what} failed with code {rc}: {msg}")


def f32x(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])
