import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from roboticattack_amd import benchmarks, ops
dev = torch.device("cuda:0")
for B in (4, 32, 128):
    rs = np.random.RandomState(B)
    sizes = np.array([[max(1, int(100 * s))] * 2 for s in rs.uniform(0.61, 1.39, B)], np.int32)
    pdesc_n, total = ops.make_pdesc(sizes)
    pdesc = torch.from_numpy(pdesc_n).to(dev)
    base = torch.rand(3, 100, 100, device=dev)
    g = torch.randn(total, device=dev)
    f = lambda: ops.patch_resize_fwd(base, pdesc, total)
    b = lambda: ops.patch_resize_bwd(g, pdesc, 100, 100)
    print(os.path.basename(os.environ.get("VAA_LIB_PATH", "default")), "B", B, "fwd %.1f us  bwd %.1f us" % (benchmarks._time(f, 20)[0] * 1e6, benchmarks._time(b, 20)[0] * 1e6))
