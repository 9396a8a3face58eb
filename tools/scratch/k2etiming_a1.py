import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["VAA_LIB_PATH"] = os.path.join(os.getcwd(), "tools/scratch/libvaa_K2TIMING_A1.so")  # tools/scratch/build_variant.sh K2TIMING -DVAA_K2_TIMING
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
L = _lib.lib()
L.vaa_k2_set_debug.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
B, D0, D1 = 64, 1024, 1152
img = torch.from_numpy(synthetic.synth_images(1, B, "noise")).to(dev)
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
_, keep = ops.patch_apply_fwd(img, patch, xy, th, True, want_keep=True)
dy0 = (torch.randn(B, 256, D0, device=dev) * 0.1).to(torch.bfloat16)
dy1 = (torch.randn(B, 256, D1, device=dev) * 0.1).to(torch.bfloat16)
wt0 = (torch.randn(588, D0, device=dev) * 0.05).to(torch.bfloat16)
wt1 = (torch.randn(588, D1, device=dev) * 0.05).to(torch.bfloat16)
nwg = 64 * 3
dbg = torch.zeros(nwg * 16 * 6, dtype=torch.int64, device=dev)
assert L.vaa_k2_set_debug(dbg.data_ptr()) == 0
for _ in range(3):
    ops.patch_embed_grad_gather(dy0, dy1, wt0, wt1, patch, xy, th, keep, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg, 16, 6)[:, :8].astype(np.float64) / 100.0
names = ["flags", "stage0", "kloop0", "stage1", "kloop1", "store"]
print("per-wave mean us:", {n: round(float(d[:, :, i].mean()), 2) for i, n in enumerate(names)}, "total", round(float(d.sum(-1).mean()), 2), "max", round(float(d.sum(-1).max()), 2))
