"""Per-phase wall clock of K1's two workgroup roles (needs the -DVAA_K1_TIMING build: tools/scratch/libvaa_K1TIMING.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
os.environ["VAA_LIB_PATH"] = os.path.join(os.getcwd(), "tools/scratch/libvaa_K1TIMING.so")
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
L = _lib.lib()
L.vaa_k1_set_debug.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
B = 64
img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev)
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
nwg = 1024 + 784
dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
assert L.vaa_k1_set_debug(dbg.data_ptr()) == 0
for _ in range(5):
    ops.patch_apply_fwd(img, patch, xy, th, True)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
t0 = d[:, :, 7][d[:, :, 7] > 0].min()
for role, name, labels in ((1, "footprint", ["params+rowtable", "loads issued+waited", "lut", "compute+stores+keep"]), (0, "background", ["rowtable(20 rows)", "barrier", "lut+stores"])):
    m = d[:, :, 6] == role
    m &= d[:, :, 7] > 0
    sel = d[m]
    tot = sel[:, :6].sum(1) / 100.0
    start = (sel[:, 7] - t0) / 100.0
    print(name, "waves", len(sel), {l: round(float(sel[:, i].mean()) / 100.0, 2) for i, l in enumerate(labels)}, "total mean", round(float(tot.mean()), 2), "max", round(float(tot.max()), 2),
          "start mean/max us", round(float(start.mean()), 2), round(float(start.max()), 2), "end max us", round(float((start + tot).max()), 2))
# concurrency profile: waves active over time
allm = d[:, :, 7] > 0
st = (d[:, :, 7][allm] - t0) / 100.0
en = st + d[:, :, :6].sum(2)[allm] / 100.0
for t in (0.2, 0.5, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
    print("t=%.1f us active waves %d started %d finished %d" % (t, int(((st <= t) & (en > t)).sum()), int((st <= t).sum()), int((en <= t).sum())))
bgm = (d[:, :, 6] == 0) & allm
print("bg start percentiles", np.percentile((d[:, :, 7][bgm] - t0) / 100.0, [1, 10, 25, 50, 75, 90, 99]))
