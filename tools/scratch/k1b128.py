import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import c_oracle
from roboticattack_amd import benchmarks, ops, synthetic
dev = torch.device("cuda:0")
for B in (96, 128, 200):
    imgs = synthetic.synth_images(1234, B, "noise")
    img = torch.from_numpy(imgs).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    o1, k1 = ops.patch_apply_fwd(img, patch, xy, th, True)
    o2, k2 = ops.patch_apply_fwd(img, patch, xy, th, True)
    torch.cuda.synchronize()
    _, ob, ok = c_oracle.patch_apply_fwd(imgs, patch.cpu().numpy(), xy_n, th_n.reshape(B,2,3), 1, 0)
    ku = np.unpackbits(k1.cpu().numpy(), axis=-1, bitorder="little")
    print(B, "run-to-run keep mism", int((k1 != k2).sum()), "vs oracle keep mism", int((ku != ok).sum()), "out mism", int((o1.view(torch.int16).cpu().numpy().view(np.uint16) != ob).sum()))
    if B == 96:
        got = o1.view(torch.int16).cpu().numpy().view(np.uint16)
        mm = np.argwhere(got != ob)
        print("images", np.unique(mm[:,0]), "channels", np.unique(mm[:,1]), "rows", mm[:,2].min(), mm[:,2].max(), "cols", mm[:,3].min(), mm[:,3].max())
        b0 = mm[0,0]; print("xy", xy_n[b0], "theta", th_n[b0])
        sub = mm[mm[:,0]==b0]; print("count img", len(sub), "rows", np.unique(sub[:,2])[:20], "cols", np.unique(sub[:,3])[:30])
        print(got[tuple(mm[0])], ob[tuple(mm[0])])
