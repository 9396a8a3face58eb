"""Per-phase wall clock of the cross-image K2' tile GEMM (embed_dgrad_rows_kernel), per wave; needs tools/scratch/build_variant.sh K2TIMING -DVAA_K2_TIMING"""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["VAA_LIB_PATH"] = os.path.join(os.getcwd(), "tools/scratch/libvaa_K2TIMING.so")
os.environ["VAA_K2E_ROWS"] = "1"
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
L = _lib.lib()
L.vaa_k2_set_debug.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
D0, D1 = 1024, 1152
wt0 = ops.pack_embed_weights((torch.randn(588, D0, device=dev) * 0.05).to(torch.bfloat16))
wt1 = ops.pack_embed_weights((torch.randn(588, D1, device=dev) * 0.05).to(torch.bfloat16))
for B in [int(v) for v in sys.argv[1:]] or [64, 8]:
    img = torch.from_numpy(synthetic.synth_images(1, B, "noise")).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    _, _, keep_t, flags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
    dy0 = (torch.randn(B, 256, D0, device=dev) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, D1, device=dev) * 0.1).to(torch.bfloat16)
    trows = ops.tile_rows_build(flags)
    nwg = 256
    dbg = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=dev)
    assert L.vaa_k2_set_debug(dbg.data_ptr()) == 0
    for _ in range(3):
        dbg.zero_()
        ops.patch_embed_grad_gather_tiles(dy0, dy1, wt0, wt1, patch, xy, th, keep_t, flags, True, defer_reduce=True, tile_rows=trows)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(nwg, 16, 8)[:, :8].astype(np.float64) / 100.0
    busy = d[:, :, 1:].sum(-1).max(-1) > 0  # workgroups that had an item
    names = ["rows", "requests", "piece0", "kloop0", "kloop1", "kloop2", "store", "-"]
    print(f"B={B}: {int(busy.sum())} of {nwg} workgroups had an item")
    print("  per-wave mean us (busy WGs):", {n: round(float(d[busy][:, :, i].mean()), 2) for i, n in enumerate(names[:7])},
          "total", round(float(d[busy].sum(-1).mean()), 2), "max", round(float(d[busy].sum(-1).max()), 2))
    print("  idle WGs prefix:", round(float(d[~busy][:, :, 0].mean()), 2) if (~busy).any() else None)
