"""K2 accuracy against the exact (fp64-accumulated) sum of the reference's fp32 products, next to the fp32 scan-order oracle's own error."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import c_oracle
from roboticattack_amd import benchmarks, ops, synthetic
dev = "cuda:0"
for B, seed in ((64, 1), (64, 2), (8, 3), (256, 4)):
    rs = np.random.RandomState(seed)
    patch = rs.rand(3, 50, 50).astype(np.float32)
    xy, th = benchmarks.random_params(B, 50, 50, 40 + seed)
    theta = th.reshape(B, 2, 3)
    imgs = torch.zeros((B, 224, 224, 3), dtype=torch.uint8, device=dev)
    _, keep = ops.patch_apply_fwd(imgs, torch.from_numpy(patch).to(dev), torch.from_numpy(xy).to(dev), torch.from_numpy(th).to(dev), True)
    g = synthetic.synth_upstream_grad(seed, min(B, 64)).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
    gb = g.view(torch.int16).numpy().view(np.uint16)
    ex = c_oracle.patch_grad(gb, patch, xy, theta, 1, 0, f64=True)
    o32 = c_oracle.patch_grad(gb, patch, xy, theta, 1, 0)
    got = ops.patch_grad_gather(g.to(dev), torch.from_numpy(patch).to(dev), torch.from_numpy(xy).to(dev), torch.from_numpy(th).to(dev), keep, True).cpu().numpy()
    s = np.abs(ex).max()
    print(f"B={B}: HIP vs exact {np.abs(got-ex).max()/s:.2e} (rms {np.sqrt(((got-ex)**2).mean())/s:.2e}); fp32 oracle vs exact {np.abs(o32-ex).max()/s:.2e}; HIP vs fp32 oracle {np.abs(got-o32).max()/s:.2e}")
