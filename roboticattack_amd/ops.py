"""PyTorch-ROCm front end of the C-ABI (include/vaa.h): device memory, streams and autograd plumbing only.

Every function here enqueues hand-written HIP kernels from libvaa_hip.so on torch's current stream and
returns without synchronising. No function has a CPU path; tensors must live on a ROCm device.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import (  # noqa: F401  (re-exported)
    GRAD_FULL,
    GRAD_SLICE,
    LAYOUT_FULL,
    LAYOUT_ROWS,
    LOSS_CE,
    LOSS_UADA,
    LOSS_UADA_DDP,
    LOSS_UPA,
    MASK_LT_M20,
    MASK_NE_M100,
    OPT_ADAMW_HF,
    OPT_PGD_SIGN,
)
from .constants import IMG, MEAN6, STD6

_MEAN = _lib.f32x(MEAN6)
_STD = _lib.f32x(STD6)
_ws_cache: dict = {}

# Optional kernel timing (bench.py): when TIMER is a list, every hand-written kernel launch sequence is bracketed by
# torch.cuda.Event pairs recorded on the launching (= torch current) stream; entries are (name, start, end, info).
TIMER = None


class _timed:
    def __init__(self, name, **info):
        self.name, self.info = name, info

    def __enter__(self):
        if TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if TIMER is not None:
            self.e.record()
            TIMER.append((self.name, self.s, self.e, self.info))
        return False


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, dtype, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VaaError(f"{name}: expected a tensor on a ROCm device (there is no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.VaaError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.VaaError(f"{name}: expected a contiguous tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.VaaError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _workspace(device, nbytes: int, kind: str = "k2") -> torch.Tensor:
    """Scratch for one operator KIND on the current stream: K2's partial tiles, K3's row statistics and the resize adjoint each have
    their own buffer per (device, stream), so a caller that overlaps them on different streams never shares scratch (vaa.h: the
    library is re-entrant per stream, the workspace is the caller's)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, kind)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def prof_start(capacity: int = 4096) -> None:
    """Arm the library's per-dispatch timer (vaa_prof_start): every kernel it launches from now on carries its own start/stop events."""
    _lib.check(_lib.lib().vaa_prof_start(int(capacity)), "vaa_prof_start")


def prof_collect() -> list:
    """Disarm and return [(kernel name, microseconds)] in launch order (waits for the recorded dispatches)."""
    import ctypes as C

    L = _lib.lib()
    n = L.vaa_prof_stop()
    out = []
    name, us = C.c_char_p(), C.c_float()
    for i in range(n):
        _lib.check(L.vaa_prof_get(i, C.byref(name), C.byref(us)), "vaa_prof_get")
        out.append((name.value.decode(), float(us.value)))
    return out


def device_check() -> None:
    _lib.check(_lib.lib().vaa_device_check(), "vaa_device_check")


def async_error_check() -> None:
    """Raises VaaError when a kernel of this process recorded a device-side failure since the last poll (include/vaa.h: vaa_async_error).
    Call it behind a synchronisation point — the word is written by the kernel that failed."""
    _lib.check(_lib.lib().vaa_async_error(), "vaa_async_error")


# ------------------------------------------------------------------------------------------------------
# K1 / K2
# ------------------------------------------------------------------------------------------------------
def patch_apply_fwd(img_u8, patch, xy, theta, geometry: bool, mask_mode: int = MASK_LT_M20, want_keep: bool = True,
                    mean6=None, std6=None):
    """K1. img_u8 [B,224,224,3] u8, patch [3,ph,pw] f32, xy [B,2] i32, theta [B,6] f32 -> (bf16 [B,6,224,224], keep bits)."""
    B = img_u8.shape[0]
    _need(img_u8, torch.uint8, "img_u8", (B, IMG, IMG, 3))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    ph, pw = int(patch.shape[1]), int(patch.shape[2])
    out = torch.empty((B, 6, IMG, IMG), dtype=torch.bfloat16, device=img_u8.device)
    keep = torch.empty((B, 3, IMG * IMG // 8), dtype=torch.uint8, device=img_u8.device) if want_keep else None
    mean_c = _MEAN if mean6 is None else _lib.f32x(mean6)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K1_patch_apply_fwd", B=B, ph=ph, pw=pw):
        rc = _lib.lib().vaa_patch_apply_fwd(
            img_u8.data_ptr(), patch.data_ptr(), xy.data_ptr(), theta.data_ptr() if geometry else None, B, ph, pw,
            int(bool(geometry)), int(mask_mode), mean_c, std_c, out.data_ptr(), keep.data_ptr() if want_keep else None, _stream())
    _lib.check(rc, "vaa_patch_apply_fwd")
    return out, keep


def patch_apply_fwd_tiles(img_u8, patch, xy, theta, geometry: bool, mask_mode: int = MASK_LT_M20, mean6=None, std6=None, pdesc=None, max_hw=None):
    """K1 in tile-major form (vaa_patch_apply_fwd_tiles): -> (out0, out1 bf16 [B,256,588], keep_tiles u16 [B,3,256,14], tile_flags u32 [B,256]).
    out_k are the operands of the two ViT patch-embed GEMMs (tile t = ty*16 + tx, element c*196 + y*14 + x). pdesc/max_hw: per-image patches."""
    B = img_u8.shape[0]
    _need(img_u8, torch.uint8, "img_u8", (B, IMG, IMG, 3))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    if pdesc is not None:
        _need(pdesc, torch.int32, "pdesc", (B, 4))
        ph, pw = int(max_hw[0]), int(max_hw[1])
    else:
        ph, pw = int(patch.shape[1]), int(patch.shape[2])
    dev = img_u8.device
    out0 = torch.empty((B, 256, 588), dtype=torch.bfloat16, device=dev)
    out1 = torch.empty((B, 256, 588), dtype=torch.bfloat16, device=dev)
    keep_t = torch.empty((B, 3, 256, 14), dtype=torch.int16, device=dev)
    flags = torch.empty((B, 256), dtype=torch.int32, device=dev)
    mean_c = _MEAN if mean6 is None else _lib.f32x(mean6)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K1_patch_apply_fwd_tiles", B=B, ph=ph, pw=pw):
        rc = _lib.lib().vaa_patch_apply_fwd_tiles(
            img_u8.data_ptr(), patch.data_ptr(), pdesc.data_ptr() if pdesc is not None else None, xy.data_ptr(), theta.data_ptr() if geometry else None,
            B, ph, pw, int(bool(geometry)), int(mask_mode), mean_c, std_c, out0.data_ptr(), out1.data_ptr(), keep_t.data_ptr(), flags.data_ptr(), _stream())
    _lib.check(rc, "vaa_patch_apply_fwd_tiles")
    return out0, out1, keep_t, flags


def patch_grad_gather(gout_bf16, patch, xy, theta, keep_bits, geometry: bool, mask_mode: int = MASK_LT_M20, std6=None, defer_reduce: bool = False):
    """K2. gout_bf16 [B,6,224,224] bf16 -> dL/d patch [3,ph,pw] f32 (sum over the batch). defer_reduce=True returns the partial tiles
    [parts, 3*ph*pw] (a view of the workspace) for ops.step_epilogue to add."""
    B = gout_bf16.shape[0]
    _need(gout_bf16, torch.bfloat16, "gout_bf16", (B, 6, IMG, IMG))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    if keep_bits is not None:
        _need(keep_bits, torch.uint8, "keep_bits", (B, 3, IMG * IMG // 8))
    ph, pw = int(patch.shape[1]), int(patch.shape[2])
    L = _lib.lib()
    nbytes = L.vaa_patch_grad_ws_bytes(B, ph, pw)
    ws = _workspace(patch.device, nbytes)
    gpatch = None if defer_reduce else torch.empty_like(patch)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_grad_gather", B=B, ph=ph, pw=pw):
        rc = L.vaa_patch_grad_gather(
            gout_bf16.data_ptr(), patch.data_ptr(), xy.data_ptr(), theta.data_ptr() if geometry else None,
            keep_bits.data_ptr() if keep_bits is not None else None, B, ph, pw, int(bool(geometry)), int(mask_mode), std_c,
            gpatch.data_ptr() if gpatch is not None else None, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_grad_gather")
    if defer_reduce:
        parts, n = L.vaa_patch_grad_partials(B), 3 * ph * pw
        return ws[: parts * n * 4].view(torch.float32).view(parts, n)
    return gpatch


def pack_embed_weights(wt):
    """Conv weights of one tower, flattened and transposed to wt [588,D] bf16, re-ordered into the MFMA fragment order K2' reads
    (vaa_patch_embed_pack_weights; once per model — the weights are frozen during an attack). Returns bf16 [592*D]."""
    D = int(wt.shape[1])
    _need(wt, torch.bfloat16, "wt", (588, D))
    L = _lib.lib()
    n = L.vaa_patch_embed_packed_elems(D)
    if n == 0:
        raise ValueError(f"pack_embed_weights: tower width {D} is not a multiple of 64")
    packed = torch.empty(n, dtype=torch.bfloat16, device=wt.device)
    _lib.check(L.vaa_patch_embed_pack_weights(wt.data_ptr(), D, packed.data_ptr(), _stream()), "vaa_patch_embed_pack_weights")
    return packed


def patch_embed_grad_gather(dy0, dy1, wp0, wp1, patch, xy, theta, keep_bits, geometry: bool, mask_mode: int = MASK_LT_M20, std6=None,
                            round_bf16: bool = True):
    """K2' (SURVEY.md 8f-3): dL/d patch from the gradients of the two ViT patch-embed OUTPUTS. dy0 [B,256,D0], dy1 [B,256,D1] bf16
    (tokens in tile order), wp0 / wp1 = pack_embed_weights(W^T [588,D]) of the two towers. Only tiles with kept pixels are
    evaluated (MFMA); the pixel gradient is never materialised."""
    B = dy0.shape[0]
    D0, D1 = int(dy0.shape[2]), int(dy1.shape[2])
    _need(dy0, torch.bfloat16, "dy0", (B, 256, D0))
    _need(dy1, torch.bfloat16, "dy1", (B, 256, D1))
    _need(wp0, torch.bfloat16, "wp0", (592 * D0,))
    _need(wp1, torch.bfloat16, "wp1", (592 * D1,))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    _need(keep_bits, torch.uint8, "keep_bits", (B, 3, IMG * IMG // 8))
    ph, pw = int(patch.shape[1]), int(patch.shape[2])
    L = _lib.lib()
    ws = _workspace(patch.device, L.vaa_patch_embed_grad_ws_bytes(B, ph, pw), "k2e")
    gpatch = torch.empty_like(patch)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_embed_grad_gather", B=B, ph=ph, pw=pw):
        rc = L.vaa_patch_embed_grad_gather(dy0.data_ptr(), D0, dy1.data_ptr(), D1, wp0.data_ptr(), wp1.data_ptr(), patch.data_ptr(), xy.data_ptr(),
                                           theta.data_ptr() if geometry else None, keep_bits.data_ptr(), B, ph, pw, int(bool(geometry)),
                                           int(mask_mode), std_c, int(bool(round_bf16)), gpatch.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_embed_grad_gather")
    return gpatch


def patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, theta, keep_tiles, tile_flags, geometry: bool, mask_mode: int = MASK_LT_M20,
                                  std6=None, round_bf16: bool = True, defer_reduce: bool = False):
    """K2' fed by the tile-major mask of patch_apply_fwd_tiles. defer_reduce=True returns (partials view [parts, 3*ph*pw] of the workspace):
    the fixed-order sum is then left to ops.step_epilogue."""
    B = dy0.shape[0]
    D0, D1 = int(dy0.shape[2]), int(dy1.shape[2])
    _need(dy0, torch.bfloat16, "dy0", (B, 256, D0))
    _need(dy1, torch.bfloat16, "dy1", (B, 256, D1))
    _need(wp0, torch.bfloat16, "wp0", (592 * D0,))
    _need(wp1, torch.bfloat16, "wp1", (592 * D1,))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    _need(keep_tiles, torch.int16, "keep_tiles", (B, 3, 256, 14))
    _need(tile_flags, torch.int32, "tile_flags", (B, 256))
    ph, pw = int(patch.shape[1]), int(patch.shape[2])
    L = _lib.lib()
    ws = _workspace(patch.device, L.vaa_patch_embed_grad_ws_bytes(B, ph, pw), "k2e")
    gpatch = None if defer_reduce else torch.empty_like(patch)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_embed_grad_gather_tiles", B=B, ph=ph, pw=pw):
        rc = L.vaa_patch_embed_grad_gather_tiles(dy0.data_ptr(), D0, dy1.data_ptr(), D1, wp0.data_ptr(), wp1.data_ptr(), patch.data_ptr(), xy.data_ptr(),
                                                 theta.data_ptr() if geometry else None, keep_tiles.data_ptr(), tile_flags.data_ptr(), B, ph, pw,
                                                 int(bool(geometry)), int(mask_mode), std_c, int(bool(round_bf16)),
                                                 gpatch.data_ptr() if gpatch is not None else None, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_embed_grad_gather_tiles")
    if defer_reduce:
        parts, n = L.vaa_patch_grad_partials(B), 3 * ph * pw
        return ws[: parts * n * 4].view(torch.float32).view(parts, n)
    return gpatch


def patch_embed_grad_gather_multi(dy0, dy1, wp0, wp1, packed, pdesc, max_hw, xy, theta, keep_bits, geometry: bool, mask_mode: int = MASK_LT_M20,
                                  std6=None, round_bf16: bool = True):
    """K2' with one patch per image (resize_patch=True): like patch_grad_gather_multi, fed by the patch-embed output gradients."""
    B = dy0.shape[0]
    D0, D1 = int(dy0.shape[2]), int(dy1.shape[2])
    _need(dy0, torch.bfloat16, "dy0", (B, 256, D0))
    _need(dy1, torch.bfloat16, "dy1", (B, 256, D1))
    _need(wp0, torch.bfloat16, "wp0", (592 * D0,))
    _need(wp1, torch.bfloat16, "wp1", (592 * D1,))
    _need(packed, torch.float32, "packed")
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    _need(keep_bits, torch.uint8, "keep_bits", (B, 3, IMG * IMG // 8))
    L = _lib.lib()
    ws = _workspace(packed.device, L.vaa_patch_embed_grad_multi_ws_bytes(B), "k2e")
    gpacked = torch.zeros_like(packed)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_embed_grad_gather_multi", B=B):
        rc = L.vaa_patch_embed_grad_gather_multi(dy0.data_ptr(), D0, dy1.data_ptr(), D1, wp0.data_ptr(), wp1.data_ptr(), packed.data_ptr(), pdesc.data_ptr(),
                                                 xy.data_ptr(), theta.data_ptr() if geometry else None, keep_bits.data_ptr(), B, int(max_hw[0]), int(max_hw[1]),
                                                 int(bool(geometry)), int(mask_mode), std_c, int(bool(round_bf16)), gpacked.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_embed_grad_gather_multi")
    return gpacked


def patch_embed_grad_gather_multi_tiles(dy0, dy1, wp0, wp1, packed, pdesc, max_hw, xy, theta, keep_tiles, tile_flags, geometry: bool,
                                        mask_mode: int = MASK_LT_M20, std6=None, round_bf16: bool = True):
    """K2' with one patch per image, fed by the tile-major mask of patch_apply_fwd_tiles(pdesc=...)."""
    B = dy0.shape[0]
    D0, D1 = int(dy0.shape[2]), int(dy1.shape[2])
    _need(dy0, torch.bfloat16, "dy0", (B, 256, D0))
    _need(dy1, torch.bfloat16, "dy1", (B, 256, D1))
    _need(wp0, torch.bfloat16, "wp0", (592 * D0,))
    _need(wp1, torch.bfloat16, "wp1", (592 * D1,))
    _need(packed, torch.float32, "packed")
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    _need(keep_tiles, torch.int16, "keep_tiles", (B, 3, 256, 14))
    _need(tile_flags, torch.int32, "tile_flags", (B, 256))
    L = _lib.lib()
    ws = _workspace(packed.device, L.vaa_patch_embed_grad_multi_ws_bytes(B), "k2e")
    gpacked = torch.zeros_like(packed)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_embed_grad_gather_multi_tiles", B=B):
        rc = L.vaa_patch_embed_grad_gather_multi_tiles(dy0.data_ptr(), D0, dy1.data_ptr(), D1, wp0.data_ptr(), wp1.data_ptr(), packed.data_ptr(), pdesc.data_ptr(),
                                                       xy.data_ptr(), theta.data_ptr() if geometry else None, keep_tiles.data_ptr(), tile_flags.data_ptr(), B,
                                                       int(max_hw[0]), int(max_hw[1]), int(bool(geometry)), int(mask_mode), std_c, int(bool(round_bf16)),
                                                       gpacked.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_embed_grad_gather_multi_tiles")
    return gpacked


class PatchApply(torch.autograd.Function):
    """Differentiable (w.r.t. `patch`) K1: PyTorch-ROCm autograd hands the model's bf16 pixel gradient to K2."""

    @staticmethod
    def forward(ctx, patch, img_u8, xy, theta, geometry, mask_mode, mean6=None, std6=None, sink=None):
        """sink (dict, optional): as in PatchApplyEmbed — the backward leaves K2's partial tiles in sink["partials"] and no patch gradient."""
        p = patch.detach().contiguous()
        out, keep = patch_apply_fwd(img_u8, p, xy, theta, geometry, mask_mode, want_keep=True, mean6=mean6, std6=std6)
        ctx.save_for_backward(p, xy, theta if geometry else xy, keep)
        ctx.geometry, ctx.mask_mode, ctx.std6, ctx.sink = bool(geometry), int(mask_mode), std6, sink
        return out

    @staticmethod
    def backward(ctx, gout):
        patch, xy, theta, keep = ctx.saved_tensors
        g = patch_grad_gather(gout.to(torch.bfloat16).contiguous(), patch, xy, theta if ctx.geometry else None, keep,
                              ctx.geometry, ctx.mask_mode, std6=ctx.std6, defer_reduce=ctx.sink is not None)
        if ctx.sink is not None:
            ctx.sink["partials"] = g
            g = None
        return g, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# resize_patch=True (config 5): per-image patches
# ------------------------------------------------------------------------------------------------------
def make_pdesc(sizes, align: int = 4):
    """sizes [B,2] (h,w) host ints -> (pdesc int32 numpy [B,4] = {h, w, offset, 0}, packed length in floats)."""
    import numpy as np

    sizes = np.asarray(sizes, np.int32).reshape(-1, 2)
    pdesc = np.zeros((sizes.shape[0], 4), np.int32)
    off = 0
    for b, (h, w) in enumerate(sizes):
        if h <= 0 or w <= 0 or h > IMG or w > IMG:
            raise _lib.VaaError(f"resized patch {h}x{w} of image {b} does not fit the {IMG}x{IMG} frame")
        pdesc[b] = (h, w, off, 0)
        off += (3 * int(h) * int(w) + align - 1) // align * align
    return pdesc, off


def patch_resize_fwd(patch, pdesc, total: int):
    """Base patch [3,ph,pw] f32 -> packed f32 [total]: image b's antialias-bilinear resized patch [3,h_b,w_b] at pdesc[b].offset."""
    _need(patch, torch.float32, "patch")
    B = int(pdesc.shape[0])
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    packed = torch.zeros(int(total), dtype=torch.float32, device=patch.device)
    with _timed("K0_patch_resize_fwd", B=B):
        rc = _lib.lib().vaa_patch_resize_fwd(patch.data_ptr(), int(patch.shape[1]), int(patch.shape[2]), pdesc.data_ptr(), B, packed.data_ptr(), _stream())
    _lib.check(rc, "vaa_patch_resize_fwd")
    return packed


def patch_resize_bwd(gpacked, pdesc, ph: int, pw: int):
    """Adjoint of patch_resize_fwd summed over the images: gpacked f32 [total] -> d L / d base patch [3,ph,pw]."""
    _need(gpacked, torch.float32, "gpacked")
    B = int(pdesc.shape[0])
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    L = _lib.lib()
    ws = _workspace(gpacked.device, L.vaa_patch_resize_ws_bytes(B, ph, pw), "resize")
    g = torch.empty((3, ph, pw), dtype=torch.float32, device=gpacked.device)
    with _timed("K0_patch_resize_bwd", B=B):
        rc = L.vaa_patch_resize_bwd(gpacked.data_ptr(), ph, pw, pdesc.data_ptr(), B, g.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_patch_resize_bwd")
    return g


def patch_apply_fwd_multi(img_u8, packed, pdesc, max_hw, xy, theta, geometry: bool, mask_mode: int = MASK_LT_M20, want_keep: bool = True,
                          mean6=None, std6=None):
    """K1 with one patch per image (packed/pdesc from patch_resize_fwd). max_hw = (max h, max w) over the batch (host ints)."""
    B = img_u8.shape[0]
    _need(img_u8, torch.uint8, "img_u8", (B, IMG, IMG, 3))
    _need(packed, torch.float32, "packed")
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    out = torch.empty((B, 6, IMG, IMG), dtype=torch.bfloat16, device=img_u8.device)
    keep = torch.empty((B, 3, IMG * IMG // 8), dtype=torch.uint8, device=img_u8.device) if want_keep else None
    mean_c = _MEAN if mean6 is None else _lib.f32x(mean6)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K1_patch_apply_fwd_multi", B=B, ph=int(max_hw[0]), pw=int(max_hw[1])):
        rc = _lib.lib().vaa_patch_apply_fwd_multi(
            img_u8.data_ptr(), packed.data_ptr(), pdesc.data_ptr(), xy.data_ptr(), theta.data_ptr() if geometry else None, B,
            int(max_hw[0]), int(max_hw[1]), int(bool(geometry)), int(mask_mode), mean_c, std_c, out.data_ptr(),
            keep.data_ptr() if want_keep else None, _stream())
    _lib.check(rc, "vaa_patch_apply_fwd_multi")
    return out, keep


def patch_grad_gather_multi(gout_bf16, packed, pdesc, max_hw, xy, theta, keep_bits, geometry: bool, mask_mode: int = MASK_LT_M20, std6=None):
    """K2 with one patch per image: returns gpacked (layout of `packed`): d L / d (every image's own patch)."""
    B = gout_bf16.shape[0]
    _need(gout_bf16, torch.bfloat16, "gout_bf16", (B, 6, IMG, IMG))
    _need(packed, torch.float32, "packed")
    _need(pdesc, torch.int32, "pdesc", (B, 4))
    _need(xy, torch.int32, "xy", (B, 2))
    if geometry:
        _need(theta, torch.float32, "theta", (B, 6))
    if keep_bits is not None:
        _need(keep_bits, torch.uint8, "keep_bits", (B, 3, IMG * IMG // 8))
    gpacked = torch.zeros_like(packed)
    std_c = _STD if std6 is None else _lib.f32x(std6)
    with _timed("K2_patch_grad_gather_multi", B=B, ph=int(max_hw[0]), pw=int(max_hw[1])):
        rc = _lib.lib().vaa_patch_grad_gather_multi(
            gout_bf16.data_ptr(), packed.data_ptr(), pdesc.data_ptr(), xy.data_ptr(), theta.data_ptr() if geometry else None,
            keep_bits.data_ptr() if keep_bits is not None else None, B, int(max_hw[0]), int(max_hw[1]), int(bool(geometry)), int(mask_mode),
            std_c, gpacked.data_ptr(), _stream())
    _lib.check(rc, "vaa_patch_grad_gather_multi")
    return gpacked


class PatchApplyResized(torch.autograd.Function):
    """resize_patch=True: resize (K0) + K1 with per-image patches forward; K2 (per-image gradients) + resize adjoint backward.
    `sizes` is a host int array [B,2] (h,w); four launches + one fixed-order reduce per step, independent of B."""

    @staticmethod
    def forward(ctx, patch, img_u8, sizes, xy, theta, geometry, mask_mode, mean6=None, std6=None):
        p = patch.detach().contiguous()
        pdesc_np, total = make_pdesc(sizes)
        pdesc = torch.from_numpy(pdesc_np).to(p.device, non_blocking=True)
        max_hw = (int(pdesc_np[:, 0].max()), int(pdesc_np[:, 1].max()))
        packed = patch_resize_fwd(p, pdesc, total)
        out, keep = patch_apply_fwd_multi(img_u8, packed, pdesc, max_hw, xy, theta, geometry, mask_mode, want_keep=True, mean6=mean6, std6=std6)
        ctx.save_for_backward(packed, pdesc, xy, theta if geometry else xy, keep)
        ctx.geometry, ctx.mask_mode, ctx.std6, ctx.max_hw = bool(geometry), int(mask_mode), std6, max_hw
        ctx.base_hw = (int(p.shape[1]), int(p.shape[2]))
        return out

    @staticmethod
    def backward(ctx, gout):
        packed, pdesc, xy, theta, keep = ctx.saved_tensors
        gp = patch_grad_gather_multi(gout.to(torch.bfloat16).contiguous(), packed, pdesc, ctx.max_hw, xy, theta if ctx.geometry else None, keep,
                                     ctx.geometry, ctx.mask_mode, std6=ctx.std6)
        g = patch_resize_bwd(gp, pdesc, *ctx.base_hw)
        return g, None, None, None, None, None, None, None, None


class PatchEmbeds(tuple):
    """(e0, e1): the two ViT patch-embed outputs [B,256,D] of a patched batch, standing in for `pixel_values` on the path that
    keeps the pixel gradient un-materialised (PatchApplyEmbed). `OpenVLAShaped.forward_rows(..., patch_embeds=...)` consumes it."""


def unfold_tiles(x3: torch.Tensor) -> torch.Tensor:
    """[B,3,224,224] -> [B,256,588]: the 14x14 tiles of timm's PatchEmbed conv (kernel == stride), columns ordered (c, y, x)."""
    B = x3.shape[0]
    return x3.reshape(B, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(B, 256, 588)


class PatchApplyEmbed(torch.autograd.Function):
    """K1 + both patch-embed GEMMs forward; backward = K2' (SURVEY.md 8f-3): the gradients of the patch-embed OUTPUTS go straight to
    `patch_embed_grad_gather`, which evaluates the patch-embed backward only for the tiles under the patch. w: [D,588], wp: pack_embed_weights(w^T)."""

    @staticmethod
    def forward(ctx, patch, img_u8, xy, theta, geometry, mask_mode, mean6, std6, w0, b0, wp0, w1, b1, wp1, sink=None):
        """sink (a dict, optional): the backward then leaves K2''s partial tiles in sink["partials"] and returns NO gradient for `patch`
        — the caller's step epilogue (ops.step_epilogue) adds them straight into the DDP message."""
        ctx.sink = sink
        p = patch.detach().contiguous()
        # K1 writes the two GEMM operands directly (tile-major): no [B,6,224,224] tensor, no im2col copies
        t0, t1, keep_t, flags = patch_apply_fwd_tiles(img_u8, p, xy, theta, geometry, mask_mode, mean6=mean6, std6=std6)
        e0 = torch.nn.functional.linear(t0, w0, b0)
        e1 = torch.nn.functional.linear(t1, w1, b1)
        ctx.save_for_backward(p, xy, theta if geometry else xy, keep_t, flags, wp0, wp1)
        ctx.geometry, ctx.mask_mode, ctx.std6 = bool(geometry), int(mask_mode), std6
        return e0, e1

    @staticmethod
    def backward(ctx, d0, d1):
        patch, xy, theta, keep_t, flags, wp0, wp1 = ctx.saved_tensors
        d0, d1 = d0.to(torch.bfloat16).contiguous(), d1.to(torch.bfloat16).contiguous()
        g = patch_embed_grad_gather_tiles(d0, d1, wp0, wp1, patch, xy, theta if ctx.geometry else None, keep_t, flags, ctx.geometry,
                                          ctx.mask_mode, std6=ctx.std6, defer_reduce=ctx.sink is not None)
        if ctx.sink is not None:
            ctx.sink["partials"] = g
            g = None
        return (g,) + (None,) * 14


class PatchApplyResizedEmbed(torch.autograd.Function):
    """resize_patch=True with the pixel gradient un-materialised: resize (K0) + K1 with per-image patches + both patch-embed GEMMs forward;
    K2' in per-image mode + the resize adjoint backward."""

    @staticmethod
    def forward(ctx, patch, img_u8, sizes, xy, theta, geometry, mask_mode, mean6, std6, w0, b0, wp0, w1, b1, wp1):
        p = patch.detach().contiguous()
        pdesc_np, total = make_pdesc(sizes)
        pdesc = torch.from_numpy(pdesc_np).to(p.device, non_blocking=True)
        max_hw = (int(pdesc_np[:, 0].max()), int(pdesc_np[:, 1].max()))
        packed = patch_resize_fwd(p, pdesc, total)
        # K1 with per-image patches, tile-major: the two GEMM operands directly (no [B,6,224,224] tensor, no im2col copies)
        t0, t1, keep_t, flags = patch_apply_fwd_tiles(img_u8, packed, xy, theta, geometry, mask_mode, mean6=mean6, std6=std6, pdesc=pdesc, max_hw=max_hw)
        e0 = torch.nn.functional.linear(t0, w0, b0)
        e1 = torch.nn.functional.linear(t1, w1, b1)
        ctx.save_for_backward(packed, pdesc, xy, theta if geometry else xy, keep_t, flags, wp0, wp1)
        ctx.geometry, ctx.mask_mode, ctx.std6, ctx.max_hw = bool(geometry), int(mask_mode), std6, max_hw
        ctx.base_hw = (int(p.shape[1]), int(p.shape[2]))
        return e0, e1

    @staticmethod
    def backward(ctx, d0, d1):
        packed, pdesc, xy, theta, keep_t, flags, wp0, wp1 = ctx.saved_tensors
        gp = patch_embed_grad_gather_multi_tiles(d0.to(torch.bfloat16).contiguous(), d1.to(torch.bfloat16).contiguous(), wp0, wp1, packed, pdesc, ctx.max_hw,
                                                 xy, theta if ctx.geometry else None, keep_t, flags, ctx.geometry, ctx.mask_mode, std6=ctx.std6)
        g = patch_resize_bwd(gp, pdesc, *ctx.base_hw)
        return (g,) + (None,) * 14


# ------------------------------------------------------------------------------------------------------
# K3
# ------------------------------------------------------------------------------------------------------
def loss_fwd_bwd(logits, labels, mode: int, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0,
                 layout: int = LAYOUT_FULL, want_grad: bool = True, want_pred: bool = True, glogits=None, want_pred_full: bool = False):
    """K3. Returns (scalars f32[8] on device, pred_tokens i32 [B,L-1] or None, glogits or None[, pred_full i32 [B,L-1]]).

    scalars = [total, CE, w^2*MSE, UPA angle, UPA dist, #CE rows, #action rows, UAD]. pred_tokens = 31744 + argmax of the action
    slice (what UAD uses, UADA.py:395); pred_full (want_pred_full) = argmax over the whole vocabulary (UADA.py:165-167 metrics)."""
    if logits.dtype == torch.float32:
        dt = _lib.DTYPE_F32
    elif logits.dtype == torch.bfloat16:
        dt = _lib.DTYPE_BF16
    else:
        raise _lib.VaaError(f"logits: unsupported dtype {logits.dtype}")
    _need(logits, logits.dtype, "logits")
    _need(labels, torch.int64, "labels")
    B, Lt = int(labels.shape[0]), int(labels.shape[1])
    if layout == LAYOUT_FULL:
        if logits.dim() != 3 or logits.shape[0] != B:
            raise _lib.VaaError(f"logits: FULL layout expects [B,S,V], got {tuple(logits.shape)}")
        S, V = int(logits.shape[1]), int(logits.shape[2])
    else:
        if logits.dim() != 2:
            raise _lib.VaaError(f"logits: ROWS layout expects [R,V], got {tuple(logits.shape)}")
        S, V = int(logits.shape[0]), int(logits.shape[1])  # ROWS: S carries the row count R
    L = _lib.lib()
    ws = _workspace(logits.device, L.vaa_loss_ws_bytes(B, Lt), "k3")
    scalars = torch.empty(8, dtype=torch.float32, device=logits.device)
    pred = torch.empty((B, Lt - 1), dtype=torch.int32, device=logits.device) if want_pred else None
    pred_full = torch.empty((B, Lt - 1), dtype=torch.int32, device=logits.device) if want_pred_full else None
    if want_grad and glogits is None:
        glogits = torch.zeros_like(logits) if layout == LAYOUT_FULL else torch.empty_like(logits)
    with _timed("K3_loss_fwd_bwd", B=B, L=Lt, V=V, dtype=str(logits.dtype), rows=(int(logits.shape[0]) if layout == LAYOUT_ROWS else -1)):
        rc = L.vaa_loss_fwd_bwd_ex(
            logits.data_ptr(), dt, int(layout), labels.data_ptr(), B, S, Lt, V, int(mode), _lib.f32x([w, alpha, beta, scale]),
            scalars.data_ptr(), pred.data_ptr() if want_pred else None, pred_full.data_ptr() if want_pred_full else None,
            glogits.data_ptr() if want_grad else None, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_loss_fwd_bwd")
    if want_pred_full:
        return scalars, pred, (glogits if want_grad else None), pred_full
    return scalars, pred, (glogits if want_grad else None)


class DiscrepancyLoss(torch.autograd.Function):
    """total = loss(logits, labels) with d total/d logits produced by the same fused launch sequence.
    Returns (total, scalars f32[8], pred_slice i32 [B,L-1], pred_full i32 [B,L-1])."""

    @staticmethod
    def forward(ctx, logits, labels, mode, w, alpha, beta, scale, layout):
        scalars, pred, g, pred_full = loss_fwd_bwd(logits.detach(), labels, mode, w, alpha, beta, scale, layout, want_grad=True,
                                                   want_pred_full=True)
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(scalars, pred, pred_full)
        return scalars[0].clone(), scalars, pred, pred_full

    @staticmethod
    def backward(ctx, gtotal, _gs, _gp, _gf):
        (g,) = ctx.saved_tensors
        return g * gtotal.to(g.dtype), None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# K3 on the labelled rows with a prebuilt row map (what the attack loops use)
# ------------------------------------------------------------------------------------------------------
ACTION_LO, N_ACTION = 31744, 256
import os as _os


SLICE_MODES = (LOSS_UADA_DDP, LOSS_UPA)  # gradient confined to the 256 action columns


class LossRowMap:
    """Device row map of a label matrix [B,L] (vaa_loss_rowmap_build): built once per outer iteration, reused by every inner step."""

    def __init__(self, labels: torch.Tensor):
        _need(labels, torch.int64, "labels")
        self.B, self.L = int(labels.shape[0]), int(labels.shape[1])
        L = _lib.lib()
        self.buf = torch.empty(L.vaa_loss_rowmap_bytes(self.B, self.L), dtype=torch.uint8, device=labels.device)
        _lib.check(L.vaa_loss_rowmap_build(labels.data_ptr(), self.B, self.L, self.buf.data_ptr(), self.buf.numel(), _stream()), "vaa_loss_rowmap_build")


def loss_rows_fwd_bwd(logits, rowmap: LossRowMap, mode: int, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0,
                      want_grad: bool = True, grad_kind: int = GRAD_FULL, want_pred: bool = True, grad=None):
    """K3 on logits [R,V] of the labelled rows. Returns (scalars f32[8], pred_slice i32 [B,L-1] | None, pred_full i32 [B,L-1] | None,
    grad [R,V] | [R,256] | None). pred_slice = 31744 + argmax of the action logits (UAD), pred_full = argmax over the vocabulary."""
    if logits.dtype == torch.float32:
        dt = _lib.DTYPE_F32
    elif logits.dtype == torch.bfloat16:
        dt = _lib.DTYPE_BF16
    else:
        raise _lib.VaaError(f"logits: unsupported dtype {logits.dtype}")
    _need(logits, logits.dtype, "logits")
    if logits.dim() != 2:
        raise _lib.VaaError(f"logits: expected [R,V], got {tuple(logits.shape)}")
    R, V = int(logits.shape[0]), int(logits.shape[1])
    B, Lt = rowmap.B, rowmap.L
    L = _lib.lib()
    ws = _workspace(logits.device, L.vaa_loss_rows_ws_bytes(R), "k3")
    scalars = torch.empty(8, dtype=torch.float32, device=logits.device)
    pred = torch.empty((B, Lt - 1), dtype=torch.int32, device=logits.device) if want_pred else None
    pred_full = torch.empty((B, Lt - 1), dtype=torch.int32, device=logits.device) if want_pred else None
    if want_grad and grad is None:
        grad = torch.empty((R, N_ACTION if grad_kind == GRAD_SLICE else V), dtype=logits.dtype, device=logits.device)
    ws = ws if R > 0 else _workspace(logits.device, 256, "k3")
    with _timed("K3_loss_rows_fwd_bwd", B=B, L=Lt, V=V, dtype=str(logits.dtype), rows=R, grad_kind=grad_kind):
        rc = L.vaa_loss_rows_fwd_bwd(logits.data_ptr(), dt, rowmap.buf.data_ptr(), R, B, Lt, V, int(mode), _lib.f32x([w, alpha, beta, scale]),
                                     scalars.data_ptr(), pred.data_ptr() if want_pred else None, pred_full.data_ptr() if want_pred else None,
                                     grad.data_ptr() if want_grad else None, int(grad_kind), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_loss_rows_fwd_bwd")
    return scalars, pred, pred_full, (grad if want_grad else None)


def loss_rows_stats(logits, rowmap: LossRowMap, mode: int, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0, grad=None):
    """The statistics pass of K3 alone (vaa_loss_rows_stats). In LOSS_UADA_DDP mode `grad` [R,256] receives the gradient slice in the same pass.
    Returns the workspace tensor that step_epilogue folds into the scalars."""
    dt = _lib.DTYPE_F32 if logits.dtype == torch.float32 else _lib.DTYPE_BF16
    _need(logits, logits.dtype, "logits")
    if logits.dim() != 2 or logits.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.VaaError(f"logits: expected f32|bf16 [R,V], got {logits.dtype} {tuple(logits.shape)}")
    R, V = int(logits.shape[0]), int(logits.shape[1])
    L = _lib.lib()
    ws = _workspace(logits.device, L.vaa_loss_rows_ws_bytes(R), "k3")
    if grad is not None:
        _need(grad, logits.dtype, "grad", (R, N_ACTION))
    with _timed("K3_loss_rows_stats", B=rowmap.B, L=rowmap.L, V=V, rows=R):
        rc = L.vaa_loss_rows_stats(logits.data_ptr(), dt, rowmap.buf.data_ptr(), R, rowmap.B, rowmap.L, V, int(mode), _lib.f32x([w, alpha, beta, scale]),
                                   grad.data_ptr() if grad is not None else None, GRAD_SLICE, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "vaa_loss_rows_stats")
    return ws


def head_loss_rows_applies(R: int, D: int, V: int) -> bool:
    return bool(_lib.lib().vaa_head_loss_rows_applies(int(R), int(D), int(V)))


def head_loss_rows_stats(hidden, w_head, rowmap: "LossRowMap", mode: int = LOSS_UADA_DDP, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2,
                         scale: float = 1.0, grad=None, want_logits: bool = False):
    """vaa_head_loss_rows_stats (see _head_stats): returns the K3 workspace for ops.step_epilogue(loss_ws=...) exactly like loss_rows_stats
    (and the bf16 logits [R,V] the statistics were made of when want_logits: tests)."""
    ws, _, dbg = _head_stats(hidden, w_head, rowmap, mode, w, alpha, beta, scale, grad, want_logits)
    return (ws, dbg) if want_logits else ws


def _head_stats(hidden, w_head, rowmap, mode, w, alpha, beta, scale, grad, want_logits):
    """vaa_head_loss_rows_stats: LM head on the labelled rows fused with K3's statistics — `hidden` [R,D] bf16, `w_head` [V,D]
    bf16; the [R,V] logits are never written. Returns (K3 workspace, the head's own workspace [per-workgroup PartStats + slice logits] — the
    very tensor vaa_head_loss_rows_finish must be handed, not a second look-up of the cache —, debug logits | None)."""
    R, D = int(hidden.shape[0]), int(hidden.shape[1])
    V = int(w_head.shape[0])
    _need(hidden, torch.bfloat16, "hidden", (R, D))
    _need(w_head, torch.bfloat16, "w_head", (V, D))
    if grad is not None:
        _need(grad, torch.bfloat16, "grad", (R, N_ACTION))
    L = _lib.lib()
    ws = _workspace(hidden.device, L.vaa_loss_rows_ws_bytes(R), "k3")
    hws = _workspace(hidden.device, L.vaa_head_loss_ws_bytes(R, V), "k3h")
    dbg = torch.empty((R, V), dtype=torch.bfloat16, device=hidden.device) if want_logits else None
    with _timed("K3_head_loss_rows_stats", rows=R, V=V, D=D):
        rc = L.vaa_head_loss_rows_stats(hidden.data_ptr(), w_head.data_ptr(), D, rowmap.buf.data_ptr(), R, rowmap.B, rowmap.L, V, int(mode),
                                        _lib.f32x([w, alpha, beta, scale]), grad.data_ptr() if grad is not None else None, ws.data_ptr(), ws.numel(),
                                        hws.data_ptr(), hws.numel(), dbg.data_ptr() if dbg is not None else None, _stream())
    _lib.check(rc, "vaa_head_loss_rows_stats")
    return ws, hws, dbg


def head_loss_rows_fwd_bwd(hidden, w_head, rowmap: "LossRowMap", mode: int, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0,
                           want_grad: bool = True, want_pred: bool = True, want_logits: bool = False):
    """LM head + K3 on the labelled rows WITHOUT logits in memory: vaa_head_loss_rows_stats (weight stream + per-row fold) followed by
    vaa_head_loss_rows_finish (scalars, prediction maps, UPA's gradient slice). Same returns as loss_rows_fwd_bwd with GRAD_SLICE:
    (scalars f32[8], pred_slice, pred_full, grad [R,256] bf16 | None). A gradient exists for the modes whose loss lives in the action
    columns (SLICE_MODES); the other modes are evaluated only (want_grad=False: validation passes)."""
    if want_grad and mode not in SLICE_MODES:
        raise _lib.VaaError(f"head_loss_rows_fwd_bwd: mode {mode} has a cross-entropy term — its gradient needs the [R,V] logits (HeadLossRows)")
    R = int(hidden.shape[0])
    V = int(w_head.shape[0])
    dev = hidden.device
    grad = torch.empty((R, N_ACTION), dtype=torch.bfloat16, device=dev) if want_grad else None
    ws, hws, lg = _head_stats(hidden, w_head, rowmap, mode, w, alpha, beta, scale, grad if mode == LOSS_UADA_DDP else None, want_logits)
    L = _lib.lib()
    scalars = torch.empty(8, dtype=torch.float32, device=dev)
    pred = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=dev) if want_pred else None
    pred_full = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=dev) if want_pred else None
    with _timed("K3_head_loss_rows_finish", rows=R, V=V):
        rc = L.vaa_head_loss_rows_finish(rowmap.buf.data_ptr(), R, rowmap.B, rowmap.L, V, int(mode), _lib.f32x([w, alpha, beta, scale]), ws.data_ptr(), ws.numel(),
                                         hws.data_ptr(), hws.numel(), scalars.data_ptr(), pred.data_ptr() if want_pred else None,
                                         pred_full.data_ptr() if want_pred else None, grad.data_ptr() if (want_grad and mode == LOSS_UPA) else None, _stream())
    _lib.check(rc, "vaa_head_loss_rows_finish")
    return (scalars, pred, pred_full, grad, lg) if want_logits else (scalars, pred, pred_full, grad)


_slice_t_cache = []  # [(weakref to the weight tensor, its _version when packed, [D,256] transposed action slice)]


def head_slice_applies(R: int, D: int, V: int) -> bool:
    return bool(_lib.lib().vaa_head_slice_applies(int(R), int(D), int(V)))


def head_slice_packed(w_head: torch.Tensor) -> torch.Tensor:
    """[D,256] bf16 transposed copy of the action rows of the LM-head weight (vaa_head_slice_pack), built once per weight TENSOR OBJECT (held by a
    weak reference: an address can be recycled by the allocator, an object cannot) and kept resident — the weights are frozen (UADA_ddp.py:50-51);
    an in-place edit of the weight (its _version) rebuilds it."""
    import weakref

    V, D = int(w_head.shape[0]), int(w_head.shape[1])
    _need(w_head, torch.bfloat16, "w_head", (V, D))
    live = [e for e in _slice_t_cache if e[0]() is not None]
    if len(live) != len(_slice_t_cache):
        _slice_t_cache[:] = live
    for ref, ver, wt in _slice_t_cache:
        if ref() is w_head and ver == w_head._version and wt.device == w_head.device:
            return wt
    wt = torch.empty((D, N_ACTION), dtype=torch.bfloat16, device=w_head.device)
    _lib.check(_lib.lib().vaa_head_slice_pack(w_head.data_ptr(), D, V, wt.data_ptr(), _stream()), "vaa_head_slice_pack")
    _slice_t_cache[:] = [e for e in _slice_t_cache if e[0]() is not w_head][-7:] + [(weakref.ref(w_head), w_head._version, wt)]
    return wt


def head_slice_fwd_bwd(hidden, w_head, rowmap: "LossRowMap", mode: int, w: float = 5.0, alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0,
                       want_dh: bool = True, want_scalars: bool = True, want_grad_slice: bool = False):
    """K3s (vaa_head_slice_fwd_bwd): slice-only LM head + statistics + gradient + head backward in one launch (modes UADA_DDP / UPA).
    Returns dict(dh [R,D] bf16 | None, ws = the K3 workspace (SliceStats + neutral parts: step_epilogue(loss_ws=...) folds it),
    scalars f32[8] | None, pred, pred_full i32 [B,L-1] | None (pred_full = -1: no full-vocabulary argmax on a slice-only step),
    grad_slice [R,256] bf16 | None)."""
    if mode not in SLICE_MODES:
        raise _lib.VaaError(f"head_slice_fwd_bwd: mode {mode} has a cross-entropy term — its loss does not live in the action columns")
    R, D = int(hidden.shape[0]), int(hidden.shape[1])
    V = int(w_head.shape[0])
    _need(hidden, torch.bfloat16, "hidden", (R, D))
    _need(w_head, torch.bfloat16, "w_head", (V, D))
    dev = hidden.device
    L = _lib.lib()
    wt = head_slice_packed(w_head) if want_dh else None
    lws = _workspace(dev, L.vaa_loss_rows_ws_bytes(R), "k3")
    zws = _workspace(dev, L.vaa_head_slice_ws_bytes(R), "k3s")
    dh = torch.empty((R, D), dtype=torch.bfloat16, device=dev) if want_dh else None
    gs = torch.empty((R, N_ACTION), dtype=torch.bfloat16, device=dev) if want_grad_slice else None
    scalars = torch.empty(8, dtype=torch.float32, device=dev) if want_scalars else None
    pred = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=dev) if want_scalars else None
    pred_full = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=dev) if want_scalars else None
    with _timed("K3s_head_slice_fwd_bwd", rows=R, D=D):
        rc = L.vaa_head_slice_fwd_bwd(hidden.data_ptr(), w_head.data_ptr(), wt.data_ptr() if wt is not None else None, D, rowmap.buf.data_ptr(), R,
                                      rowmap.B, rowmap.L, V, int(mode), _lib.f32x([w, alpha, beta, scale]), dh.data_ptr() if dh is not None else None,
                                      gs.data_ptr() if gs is not None else None, lws.data_ptr(), lws.numel(),
                                      scalars.data_ptr() if scalars is not None else None, pred.data_ptr() if pred is not None else None,
                                      pred_full.data_ptr() if pred_full is not None else None, zws.data_ptr(), zws.numel(), _stream())
    _lib.check(rc, "vaa_head_slice_fwd_bwd")
    return {"dh": dh, "ws": lws, "zs": zws, "scalars": scalars, "pred": pred, "pred_full": pred_full, "grad_slice": gs}


def step_epilogue(partials, msg, scalars, rowmap: LossRowMap = None, R: int = 0, V: int = 32064, mode: int = LOSS_UADA_DDP, w: float = 5.0,
                  alpha: float = 0.8, beta: float = 0.2, scale: float = 1.0, loss_ws=None, want_pred: bool = True, update=None):
    """vaa_step_epilogue: msg[0..n) = fixed-order sum of K2's partial tiles [parts, n]; with `rowmap` (+ the workspace loss_rows_stats left)
    K3's statistics are folded into `scalars` (f32[8], output) and the prediction maps; msg[n..n+4) = {CE, w^2*MSE, UAD, total}.
    update = dict(patch=, m=, v=, mode=, lr=, step=, beta1=, beta2=, eps=, stat_part= f64 [ceil(n/64), 2]) fuses K4 (single-GPU step:
    vaa_step_epilogue_update). Returns (pred_slice, pred_full) or (None, None)."""
    _need(partials, torch.float32, "partials")
    parts, n = int(partials.shape[0]), int(partials.shape[1])
    _need(msg, torch.float32, "msg")
    _need(scalars, torch.float32, "scalars", (8,))
    if msg.numel() < n + 4:
        raise _lib.VaaError(f"msg: needs {n + 4} floats, has {msg.numel()}")
    pred = pred_full = None
    if rowmap is not None and want_pred:
        pred = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=msg.device)
        pred_full = torch.empty((rowmap.B, rowmap.L - 1), dtype=torch.int32, device=msg.device)
    common = (partials.data_ptr(), parts, n, rowmap.buf.data_ptr() if rowmap is not None else None, int(R), rowmap.B if rowmap is not None else 0,
              rowmap.L if rowmap is not None else 0, int(V), int(mode), _lib.f32x([w, alpha, beta, scale]),
              loss_ws.data_ptr() if loss_ws is not None else None, loss_ws.numel() if loss_ws is not None else 0, scalars.data_ptr(),
              pred.data_ptr() if pred is not None else None, pred_full.data_ptr() if pred_full is not None else None, msg.data_ptr())
    with _timed("EPI_step_epilogue", n=n, parts=parts):
        if update is None:
            rc = _lib.lib().vaa_step_epilogue(*common, _stream())
        else:
            u = update
            _need(u["patch"], torch.float32, "patch")
            if u["patch"].numel() != n:
                raise _lib.VaaError(f"update: patch has {u['patch'].numel()} elements, the gradient {n}")
            sp = u.get("stat_part")
            if sp is not None:
                _need(sp, torch.float64, "stat_part", ((n + 63) // 64, 2))
            rc = _lib.lib().vaa_step_epilogue_update(*common, u["patch"].data_ptr(), u["m"].data_ptr() if u.get("m") is not None else None,
                                                     u["v"].data_ptr() if u.get("v") is not None else None, int(u["mode"]), float(u["lr"]),
                                                     float(u.get("beta1", 0.9)), float(u.get("beta2", 0.999)), float(u.get("eps", 1e-6)), int(u["step"]),
                                                     sp.data_ptr() if sp is not None else None, _stream())
    _lib.check(rc, "vaa_step_epilogue")
    return pred, pred_full


class DiscrepancyLossRows(torch.autograd.Function):
    """total = loss(logits [R,V], row map); d total / d logits by the same launch sequence (full-row gradient storage)."""

    @staticmethod
    def forward(ctx, logits, rowmap, mode, w, alpha, beta, scale):
        scalars, pred, pred_full, g = loss_rows_fwd_bwd(logits.detach(), rowmap, mode, w, alpha, beta, scale, want_grad=True, grad_kind=GRAD_FULL)
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(scalars, pred, pred_full)
        return scalars[0].clone(), scalars, pred, pred_full

    @staticmethod
    def backward(ctx, gtotal, _gs, _gp, _gf):
        (g,) = ctx.saved_tensors
        return g * gtotal.to(g.dtype), None, None, None, None, None, None


class HeadLossRows(torch.autograd.Function):
    """LM head + loss on the labelled rows: logits = hidden [R,D] @ W^T [V,D] through hipBLASLt, K3 on them, and
    for the modes whose gradient lives in the 256 action columns (UADA_DDP, UPA) the backward is dh = g_slice [R,256] @ W[31744:32000]
    — a contraction over 256 columns instead of the 32,064 of the generic head backward; other modes contract over the full rows."""

    @staticmethod
    def forward(ctx, hidden, weight, rowmap, mode, w, alpha, beta, scale):
        logits = torch.nn.functional.linear(hidden.detach(), weight)
        sliced = mode in SLICE_MODES
        scalars, pred, pred_full, g = loss_rows_fwd_bwd(logits, rowmap, mode, w, alpha, beta, scale, want_grad=True,
                                                        grad_kind=GRAD_SLICE if sliced else GRAD_FULL)
        ctx.save_for_backward(g, weight)
        ctx.sliced = sliced
        ctx.mark_non_differentiable(scalars, pred, pred_full)
        return scalars[0].clone(), scalars, pred, pred_full

    @staticmethod
    def backward(ctx, gtotal, _gs, _gp, _gf):
        g, weight = ctx.saved_tensors
        wsel = weight[ACTION_LO : ACTION_LO + N_ACTION] if ctx.sliced else weight
        dh = (g * gtotal.to(g.dtype)) @ wsel
        return dh, None, None, None, None, None, None, None


class HeadLossRowsFused(torch.autograd.Function):
    """HeadLossRows for the modes whose loss lives in the 256 action columns (UADA_DDP, UPA) with the LM head FUSED into K3's statistics
    (head_loss_rows_fwd_bwd: the [R,V] logits never reach memory); backward dh = g_slice [R,256] @ W[31744:32000] as in HeadLossRows."""

    @staticmethod
    def forward(ctx, hidden, weight, rowmap, mode, w, alpha, beta, scale):
        scalars, pred, pred_full, g = head_loss_rows_fwd_bwd(hidden.detach().contiguous(), weight, rowmap, mode, w, alpha, beta, scale, want_grad=True)
        ctx.save_for_backward(g, weight)
        ctx.mark_non_differentiable(scalars, pred, pred_full)
        return scalars[0].clone(), scalars, pred, pred_full

    @staticmethod
    def backward(ctx, gtotal, _gs, _gp, _gf):
        g, weight = ctx.saved_tensors
        dh = (g * gtotal.to(g.dtype)) @ weight[ACTION_LO : ACTION_LO + N_ACTION]
        return dh, None, None, None, None, None, None, None


_silent_cache = {}


def _silent_outputs(device, B: int, L: int):
    """What a step whose loss scalars nobody reads hands back: zeros[8] and -1 maps (allocated once per device and shape; never written)."""
    key = (device.index, B, L)
    hit = _silent_cache.get(key)
    if hit is None:
        hit = (torch.zeros(8, dtype=torch.float32, device=device), torch.full((B, L - 1), -1, dtype=torch.int32, device=device))
        _silent_cache[key] = hit
    return hit


class HeadSliceLoss(torch.autograd.Function):
    """The slice modes' head + loss + head backward as K3s (head_slice_fwd_bwd: ONE launch; d total / d hidden is produced in the forward and
    handed back by the backward). `read_scalars=False` — a step whose loss scalars the loop never reads (every inner step but the last of an
    outer iteration: UADA_ddp.py:214-221, UPA.py:171-186) — runs the launch without its fold: scalars = 0, maps = -1. `full_ce` — a step whose
    full-vocabulary CE / argmax is READ (the last inner step of the data-parallel UADA loop; validation passes that log CE) — adds K3h's
    statistics pass + fold for the scalars and prediction maps. The gradient path is K3s's on every step, so the patch trajectory does not
    depend on which steps evaluate what.
    Returns (total, scalars f32[8], pred_slice, pred_full); scalars[1] (CE) = 0 and pred_full = -1 unless full_ce."""

    @staticmethod
    def forward(ctx, hidden, weight, rowmap, mode, w, alpha, beta, scale, full_ce, read_scalars=True):
        h = hidden.detach().contiguous()
        full_ce = bool(full_ce and read_scalars)
        o = head_slice_fwd_bwd(h, weight, rowmap, mode, w, alpha, beta, scale, want_dh=True, want_scalars=bool(read_scalars and not full_ce))
        if full_ce:  # K3s left the SliceStats + neutral parts; K3h now writes the real parts (and the same SliceStat bits) into the same workspace
            scalars, pred, pred_full, _ = head_loss_rows_fwd_bwd(h, weight, rowmap, mode, w, alpha, beta, scale, want_grad=False)
        elif read_scalars:
            scalars, pred, pred_full = o["scalars"], o["pred"], o["pred_full"]
        else:
            scalars, pred = _silent_outputs(h.device, rowmap.B, rowmap.L)
            pred_full = pred
        ctx.save_for_backward(o["dh"])
        ctx.mark_non_differentiable(scalars, pred, pred_full)
        return scalars[0].clone(), scalars, pred, pred_full

    @staticmethod
    def backward(ctx, gtotal, _gs, _gp, _gf):
        (dh,) = ctx.saved_tensors
        return dh * gtotal.to(dh.dtype), None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# K4
# ------------------------------------------------------------------------------------------------------
def patch_update(patch, grad, m, v, mode: int, lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999,
                 eps: float = 1e-6, l1_clip: float = 0.0, grad_scale: float = 1.0, want_stats: bool = True):
    """K4 (in place on patch/m/v). Returns stats f32[2] = [sum|g|, mean g] (device) or None."""
    _need(patch, torch.float32, "patch")
    _need(grad, torch.float32, "grad", patch.shape)
    if mode == OPT_ADAMW_HF:
        _need(m, torch.float32, "m", patch.shape)
        _need(v, torch.float32, "v", patch.shape)
    stats = torch.empty(2, dtype=torch.float32, device=patch.device) if want_stats else None
    with _timed("K4_patch_update", n=int(patch.numel())):
        rc = _lib.lib().vaa_patch_update(
            patch.data_ptr(), grad.data_ptr(), m.data_ptr() if m is not None else None, v.data_ptr() if v is not None else None,
            patch.numel(), int(mode), float(lr), float(beta1), float(beta2), float(eps), int(step), float(l1_clip),
            float(grad_scale), stats.data_ptr() if want_stats else None, _stream())
    _lib.check(rc, "vaa_patch_update")
    return stats


# ------------------------------------------------------------------------------------------------------
# eval-time paste (simulation_random_patch)
# ------------------------------------------------------------------------------------------------------
def patch_apply_eval(img_u8, patch, xy, theta, geometry):
    """uint8 frames [B,224,224,3] + float patch [3,ph,pw] -> uint8 frames with the (uint8-quantised, optionally warped) patch."""
    B = img_u8.shape[0]
    _need(img_u8, torch.uint8, "img_u8", (B, IMG, IMG, 3))
    _need(patch, torch.float32, "patch")
    _need(xy, torch.int32, "xy", (B, 2))
    _need(theta, torch.float32, "theta", (B, 6))
    _need(geometry, torch.int32, "geometry", (B,))
    out = torch.empty_like(img_u8)
    with _timed("K5_patch_apply_eval", B=B):
        rc = _lib.lib().vaa_patch_apply_eval(img_u8.data_ptr(), patch.data_ptr(), xy.data_ptr(), theta.data_ptr(), geometry.data_ptr(), B,
                                             int(patch.shape[1]), int(patch.shape[2]), out.data_ptr(), _stream())
    _lib.check(rc, "vaa_patch_apply_eval")
    return out
