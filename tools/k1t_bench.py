#!/usr/bin/env python3
"""K1 planar vs tile-major, one batch size: python tools/k1t_bench.py <B>  (VAA_K1T_FPWGS = footprint-workgroup cap, experiment knob)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.benchmarks import _time, random_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
img = torch.from_numpy(synthetic.synth_images(1234, min(B, 64), "noise")).to(dev)
if B > 64:
    img = img.repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
a = _time(lambda: ops.patch_apply_fwd(img, patch, xy, th, True), 40)
t = _time(lambda: ops.patch_apply_fwd_tiles(img, patch, xy, th, True), 40)
print(f"cap {os.environ.get('VAA_K1T_FPWGS', '-')} B={B}: planar {a[0] * 1e6:.2f} us (min {a[2] * 1e6:.2f}), tiles {t[0] * 1e6:.2f} us (min {t[2] * 1e6:.2f})")
