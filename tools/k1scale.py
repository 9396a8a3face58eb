import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import benchmarks, ops, synthetic
dev = torch.device("cuda:0")
res = []
for geo in (True, False):
    for B in (1, 8, 16, 32, 64, 128, 256, 512):
        img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev)
        patch = torch.rand(3, 50, 50, device=dev)
        xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
        xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
        m, md, mn = benchmarks._time(lambda: ops.patch_apply_fwd(img, patch, xy, th, geo), 30)
        res.append((geo, B, round(m * 1e6, 1)))
print(res)
# reference points: a plain device copy of the same byte count, and an empty-ish op (K4)
for B in (64, 256):
    n = B * 224 * 224 * 3
    src = torch.empty(n, dtype=torch.uint8, device=dev); dst = torch.empty(n * 4, dtype=torch.uint8, device=dev)
    m, _, _ = benchmarks._time(lambda: dst.fill_(1), 30)
    print("fill", B, n * 4 / 1e6, "MB", round(m * 1e6, 1), "us")
    a = torch.empty(n * 5 // 2, dtype=torch.uint8, device=dev); b_ = torch.empty_like(a)
    m, _, _ = benchmarks._time(lambda: b_.copy_(a), 30)
    print("copy", B, n * 5 / 1e6, "MB total", round(m * 1e6, 1), "us")
