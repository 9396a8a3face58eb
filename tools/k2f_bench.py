#!/usr/bin/env python3
"""K2' as one launch (vaa_patch_embed_grad_fused) against tile GEMM + gather (vaa_patch_embed_grad_gather_tiles, final sum deferred), per-dispatch
times (vaa_prof_*):  python tools/k2f_bench.py [B ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.benchmarks import random_params  # noqa: E402

dev = torch.device("cuda:0")
for B in [int(v) for v in sys.argv[1:]] or [64, 40, 8]:
    img = torch.from_numpy(synthetic.synth_images(1234, min(B, 64), "noise")).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    dy0 = (torch.randn(B, 256, 1024, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, 1152, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    wp0 = ops.pack_embed_weights((torch.randn(588, 1024, device=dev, generator=g) * 0.05).to(torch.bfloat16))
    wp1 = ops.pack_embed_weights((torch.randn(588, 1152, device=dev, generator=g) * 0.05).to(torch.bfloat16))
    _, _, keep_t, flags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
    out = {}
    for tag, fn in (("two launches", lambda: ops.patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, th, keep_t, flags, True, defer_reduce=True)),
                    ("one launch", lambda: ops.patch_embed_grad_fused(dy0, dy1, wp0, wp1, patch.shape, xy, th, keep_t, flags, True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ops.prof_start(1024)
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        per = {}
        for n, us in ops.prof_collect():
            per.setdefault(n.split("(")[-1].split("<")[0] + ("<" + n.split("<")[1] if "<" in n else ""), []).append(us)
        out[tag] = {k[:60]: round(float(np.mean(v)), 2) for k, v in per.items()}
    print(f"B={B}:", {t: (v, "sum", round(sum(v.values()), 2)) for t, v in out.items()})
