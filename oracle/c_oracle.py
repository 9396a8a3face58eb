"""ctypes binding of oracle/libvaa_oracle.so (ORACLE — test infrastructure, never imported by the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvaa_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "vaa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libvaa_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.vaa_oracle_patch_update.restype = C.c_float
        _lib.vaa_oracle_action_argmax.restype = C.c_int
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


MEAN6 = np.array([0.484375, 0.455078125, 0.40625, 0.5, 0.5, 0.5], np.float32)
STD6 = np.array([0.228515625, 0.2236328125, 0.224609375, 0.5, 0.5, 0.5], np.float32)


def patch_apply_fwd(img_u8, patch, xy, theta, geometry, mask_mode=0, want_f32=True, want_bf16=True, want_keep=True):
    img_u8 = np.ascontiguousarray(img_u8, np.uint8)
    patch = np.ascontiguousarray(patch, np.float32)
    xy = np.ascontiguousarray(xy, np.int32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 6)
    B, ph, pw = img_u8.shape[0], patch.shape[1], patch.shape[2]
    out = np.empty((B, 6, 224, 224), np.float32) if want_f32 else None
    ob = np.empty((B, 6, 224, 224), np.uint16) if want_bf16 else None
    keep = np.empty((B, 3, 224 * 224), np.uint8) if want_keep else None
    lib().vaa_oracle_patch_apply_fwd(_p(img_u8, C.c_uint8), _p(patch, C.c_float), _p(xy, C.c_int32), _p(theta, C.c_float),
                                     B, ph, pw, int(geometry), int(mask_mode), _p(MEAN6, C.c_float), _p(STD6, C.c_float),
                                     _p(out, C.c_float), _p(ob, C.c_uint16), _p(keep, C.c_uint8))
    return out, ob, keep


def patch_grad(gout_bf16_bits, patch, xy, theta, geometry, mask_mode=0, f64=False):
    """f64=True: the same fp32 products accumulated in fp64 (the exact sum); default = fp32 scan-order accumulation like torch CPU."""
    g = np.ascontiguousarray(gout_bf16_bits, np.uint16)
    patch = np.ascontiguousarray(patch, np.float32)
    xy = np.ascontiguousarray(xy, np.int32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 6)
    B, ph, pw = g.shape[0], patch.shape[1], patch.shape[2]
    out = np.empty((3, ph, pw), np.float32)
    fn = lib().vaa_oracle_patch_grad_f64 if f64 else lib().vaa_oracle_patch_grad
    fn(_p(g, C.c_uint16), _p(patch, C.c_float), _p(xy, C.c_int32), _p(theta, C.c_float), B, ph, pw,
       int(geometry), int(mask_mode), _p(STD6, C.c_float), _p(out, C.c_float))
    return out


MODE_UADA, MODE_UADA_DDP, MODE_UPA, MODE_CE = 0, 1, 2, 3


def loss(logits, labels, mode, w=5.0, alpha=0.8, beta=0.2, scale=1.0, want_grad=True):
    logits = np.ascontiguousarray(logits, np.float32)
    labels = np.ascontiguousarray(labels, np.int64)
    B, S, V = logits.shape
    L = labels.shape[1]
    params = np.array([w, alpha, beta, scale], np.float32)
    scalars = np.zeros(8, np.float32)
    g = np.zeros_like(logits) if want_grad else None
    lib().vaa_oracle_loss(_p(logits, C.c_float), _p(labels, C.c_int64), B, S, L, V, int(mode), _p(params, C.c_float),
                          _p(scalars, C.c_float), _p(g, C.c_float))
    return scalars, g


def action_argmax(logits, labels):
    logits = np.ascontiguousarray(logits, np.float32)
    labels = np.ascontiguousarray(labels, np.int64)
    B, S, V = logits.shape
    L = labels.shape[1]
    pred = np.zeros(B * L, np.int64)
    gt = np.zeros(B * L, np.int64)
    n = lib().vaa_oracle_action_argmax(_p(logits, C.c_float), _p(labels, C.c_int64), B, S, L, V, _p(pred, C.c_int64), _p(gt, C.c_int64))
    return pred[:n], gt[:n]


def patch_update(patch, g, m, v, mode, lr, step, b1=0.9, b2=0.999, eps=1e-6, l1_clip=0.0, grad_scale=1.0):
    """In place on patch/m/v (float32 contiguous). Returns sum|g|."""
    for a in (patch, g, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    return lib().vaa_oracle_patch_update(_p(patch, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float), patch.size,
                                         int(mode), C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps), int(step),
                                         C.c_float(l1_clip), C.c_float(grad_scale))


def patch_apply_eval(img_u8, patch, xy, theta, geometry):
    """simulation_random_patch for a batch: geometry is a per-image bool array."""
    img_u8 = np.ascontiguousarray(img_u8, np.uint8)
    patch = np.ascontiguousarray(patch, np.float32)
    xy = np.ascontiguousarray(xy, np.int32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 6)
    geo = np.ascontiguousarray(np.asarray(geometry).astype(np.int32))
    out = np.empty_like(img_u8)
    lib().vaa_oracle_patch_apply_eval(_p(img_u8, C.c_uint8), _p(patch, C.c_float), _p(xy, C.c_int32), _p(theta, C.c_float), img_u8.shape[0],
                                      patch.shape[1], patch.shape[2], _p(geo, C.c_int32), _p(out, C.c_uint8))
    return out


# ---- resize_patch=True (config 5): per-image patches --------------------------------------------------------------
def make_pdesc(sizes, align=4):
    """sizes [B,2] (h,w) -> pdesc int32 [B,4] = {h, w, offset (floats, multiple of `align`), 0} and the packed length."""
    sizes = np.asarray(sizes, np.int32).reshape(-1, 2)
    pdesc = np.zeros((sizes.shape[0], 4), np.int32)
    off = 0
    for b, (h, w) in enumerate(sizes):
        pdesc[b] = (h, w, off, 0)
        off += (3 * int(h) * int(w) + align - 1) // align * align
    return pdesc, off


def patch_resize_fwd(patch, pdesc, total):
    patch = np.ascontiguousarray(patch, np.float32)
    pdesc = np.ascontiguousarray(pdesc, np.int32)
    packed = np.zeros(total, np.float32)
    lib().vaa_oracle_patch_resize_fwd(_p(patch, C.c_float), patch.shape[1], patch.shape[2], _p(pdesc, C.c_int32), pdesc.shape[0],
                                      _p(packed, C.c_float))
    return packed


def patch_resize_bwd(gpacked, pdesc, ph, pw):
    gpacked = np.ascontiguousarray(gpacked, np.float32)
    pdesc = np.ascontiguousarray(pdesc, np.int32)
    out = np.empty((3, ph, pw), np.float32)
    lib().vaa_oracle_patch_resize_bwd(_p(gpacked, C.c_float), ph, pw, _p(pdesc, C.c_int32), pdesc.shape[0], _p(out, C.c_float))
    return out


def patch_apply_fwd_multi(img_u8, packed, pdesc, xy, theta, geometry, mask_mode=0, want_f32=True, want_bf16=True, want_keep=True):
    img_u8 = np.ascontiguousarray(img_u8, np.uint8)
    packed = np.ascontiguousarray(packed, np.float32)
    pdesc = np.ascontiguousarray(pdesc, np.int32)
    xy = np.ascontiguousarray(xy, np.int32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 6)
    B = img_u8.shape[0]
    out = np.empty((B, 6, 224, 224), np.float32) if want_f32 else None
    ob = np.empty((B, 6, 224, 224), np.uint16) if want_bf16 else None
    keep = np.empty((B, 3, 224 * 224), np.uint8) if want_keep else None
    lib().vaa_oracle_patch_apply_fwd_multi(_p(img_u8, C.c_uint8), _p(packed, C.c_float), _p(pdesc, C.c_int32), _p(xy, C.c_int32),
                                           _p(theta, C.c_float), B, int(geometry), int(mask_mode), _p(MEAN6, C.c_float),
                                           _p(STD6, C.c_float), _p(out, C.c_float), _p(ob, C.c_uint16), _p(keep, C.c_uint8))
    return out, ob, keep


def patch_grad_multi(gout_bf16_bits, packed, pdesc, xy, theta, geometry, mask_mode=0):
    g = np.ascontiguousarray(gout_bf16_bits, np.uint16)
    packed = np.ascontiguousarray(packed, np.float32)
    pdesc = np.ascontiguousarray(pdesc, np.int32)
    xy = np.ascontiguousarray(xy, np.int32)
    theta = np.ascontiguousarray(theta, np.float32).reshape(-1, 6)
    out = np.zeros_like(packed)
    lib().vaa_oracle_patch_grad_multi(_p(g, C.c_uint16), _p(packed, C.c_float), _p(pdesc, C.c_int32), _p(xy, C.c_int32),
                                      _p(theta, C.c_float), g.shape[0], int(geometry), int(mask_mode), _p(STD6, C.c_float),
                                      _p(out, C.c_float))
    return out
