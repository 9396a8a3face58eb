import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import benchmarks, ops, synthetic
dev = torch.device("cuda:0")
res = []
for B in (64, 256):
    for kind in ("noise", "zeros"):
        img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev) if kind == "noise" else torch.zeros((B, 224, 224, 3), dtype=torch.uint8, device=dev)
        patch = torch.rand(3, 50, 50, device=dev)
        xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
        xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
        m, md, mn = benchmarks._time(lambda: ops.patch_apply_fwd(img, patch, xy, th, True), 30)
        res.append((B, kind, round(m * 1e6, 1), round(mn * 1e6, 1)))
print(os.environ.get("VAA_LIB_PATH", "default").split("/")[-1], res)
