import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["VAA_LIB_PATH"] = os.path.join(os.getcwd(), "tools/scratch/libvaa_K2TIMING.so")  # tools/scratch/build_variant.sh K2TIMING -DVAA_K2_TIMING
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
L = _lib.lib()
L.vaa_k2_set_debug.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
for B in (64, 4096):
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    img = torch.zeros((B, 224, 224, 3), dtype=torch.uint8, device=dev)
    _, keep = ops.patch_apply_fwd(img, patch, xy, th, True)
    g = synthetic.synth_upstream_grad(7, 64).to(dev).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
    nwg = 64 if B == 64 else 512  # blockIdx.x range (the row bands of an image, grid.z, write the same slots: one of them is reported)
    dbg = torch.zeros(nwg * 16 * 6, dtype=torch.int64, device=dev)
    assert L.vaa_k2_set_debug(dbg.data_ptr()) == 0
    for _ in range(3):
        ops.patch_grad_gather(g, patch, xy, th, keep, True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 16, 6)[:, :8].astype(np.float64) / 100.0  # 512-thread workgroups: 8 waves  # wall_clock64: 100 MHz -> us
    names = ["tables", "phase1a(issue)", "phase1b(consume)", "barrier", "phase2+rescale", "drain"]
    print("B", B, "per-wave mean us:", {n: round(float(d[:, :, i].mean()), 2) for i, n in enumerate(names)}, "total", round(float(d.sum(-1).mean()), 2), "max wave total", round(float(d.sum(-1).max()), 2))
