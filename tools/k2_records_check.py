#!/usr/bin/env python3
"""The records experiment: K1's footprint role emits per-pixel records {x0, y0, kept, w, n}; the TILED gather of K2'
walks the flagged tiles and reads them instead of rebuilding row tables + the exact coordinate chain. Prints bitwise equality of the result
and per-dispatch times (vaa_prof_*) of K1 and of the gather, with and without records.   python tools/k2_records_check.py [B ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import _lib, ops, synthetic  # noqa: E402
from roboticattack_amd.benchmarks import random_params  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
for B in [int(v) for v in sys.argv[1:]] or [64, 8]:
    img = torch.from_numpy(synthetic.synth_images(1234, min(B, 64), "noise")).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    dy0 = (torch.randn(B, 256, 1024, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, 1152, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    wp0 = ops.pack_embed_weights((torch.randn(588, 1024, device=dev, generator=g) * 0.05).to(torch.bfloat16))
    wp1 = ops.pack_embed_weights((torch.randn(588, 1152, device=dev, generator=g) * 0.05).to(torch.bfloat16))
    t0 = torch.empty((B, 256, 588), dtype=torch.bfloat16, device=dev)
    t1 = torch.empty_like(t0)
    keep_t = torch.empty((B, 3, 256, 14), dtype=torch.int16, device=dev)
    flags = torch.empty((B, 256), dtype=torch.int32, device=dev)
    rec = torch.zeros((B, 256, 196, 4), dtype=torch.int32, device=dev)
    ws = torch.empty(L.vaa_patch_embed_grad_ws_bytes(B, 50, 50), dtype=torch.uint8, device=dev)
    gp = [torch.empty_like(patch), torch.empty_like(patch)]

    def k1(records):
        args = (img.data_ptr(), patch.data_ptr(), None, xy.data_ptr(), th.data_ptr(), B, 50, 50, 1, 0, ops._MEAN, ops._STD, t0.data_ptr(), t1.data_ptr(),
                keep_t.data_ptr(), flags.data_ptr())
        rc = L.vaa_patch_apply_fwd_tiles_rec(*args, rec.data_ptr(), st()) if records else L.vaa_patch_apply_fwd_tiles(*args, st())
        _lib.check(rc, "k1")

    def k2(records):
        a = (dy0.data_ptr(), 1024, dy1.data_ptr(), 1152, wp0.data_ptr(), wp1.data_ptr(), patch.data_ptr(), xy.data_ptr(), th.data_ptr(), keep_t.data_ptr(),
             flags.data_ptr())
        b = (B, 50, 50, 1, 0, ops._STD, 1, gp[1 if records else 0].data_ptr(), ws.data_ptr(), ws.numel(), st())
        rc = L.vaa_patch_embed_grad_gather_tiles_rec(*a, rec.data_ptr(), *b) if records else L.vaa_patch_embed_grad_gather_tiles(*a, *b)
        _lib.check(rc, "k2")

    out = {}
    for records in (False, True):
        for _ in range(3):
            k1(records); k2(records)
        torch.cuda.synchronize()
        ops.prof_start(2048)
        for _ in range(30):
            k1(records); k2(records)
        torch.cuda.synchronize()
        per = {}
        for n, us in ops.prof_collect():
            per.setdefault(n.split("<")[0].strip("( "), []).append(us)
        out[records] = {n: float(np.mean(v)) for n, v in per.items()}
    same = torch.equal(gp[0], gp[1])
    rel = float((gp[0] - gp[1]).abs().max() / gp[0].abs().max())
    print(f"B={B}: records path vs product: bitwise equal {same}, max |d| / max|g| = {rel:.1e} (the slot order differs, so the workgroups' fixed-point "
          f"exponents can evolve differently: both are within 2^-30 of the exact sum)")
    for records in (False, True):
        print("   records" if records else "   product", {k: round(v, 2) for k, v in out[records].items()}, "sum", round(sum(out[records].values()), 2))
