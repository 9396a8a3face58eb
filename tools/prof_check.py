#!/usr/bin/env python3
"""Checks the library's per-dispatch timer (vaa_prof_*) against rocprofv3: run under
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python tools/prof_check.py
and compare the printed per-kernel means with the *_kernel_stats.csv averages of the same process."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.benchmarks import random_params  # noqa: E402

dev = torch.device("cuda:0")
B = 64
img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev)
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
g = synthetic.synth_upstream_grad(7, B).to(dev)
out, keep = ops.patch_apply_fwd(img, patch, xy, th, True)
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
m, v = torch.zeros_like(patch), torch.zeros_like(patch)
for _ in range(3):
    ops.patch_apply_fwd(img, patch, xy, th, True)
torch.cuda.synchronize()
ops.prof_start(1024)
for it in range(40):
    if it >= 20:
        big.fill_(it)  # cold caches for the second half
    ops.patch_apply_fwd(img, patch, xy, th, True)
    gp = ops.patch_grad_gather(g, patch, xy, th, keep, True)
    ops.patch_update(patch, gp, m, v, ops.OPT_ADAMW_HF, 1e-3, it + 1)
torch.cuda.synchronize()
recs = ops.prof_collect()
agg = {}
for name, us in recs:
    agg.setdefault(name, []).append(us)
print(json.dumps({k: {"n": len(v), "mean_us": float(np.mean(v)), "min_us": float(np.min(v)), "warm_mean_us": float(np.mean(v[: len(v) // 2])),
                      "cold_mean_us": float(np.mean(v[len(v) // 2:]))} for k, v in agg.items()}, indent=1))
