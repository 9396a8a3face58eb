"""diagnostic for one --head soak case: dumps the gradient slice of the kernel and of the oracle around the worst element"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import soak_loss as S
from oracle import c_oracle
from roboticattack_amd import ops
orig = ops.head_loss_rows_stats
cap = {}
def hook(h, W, rm, mode, w, *a, **k):
    out = orig(h, W, rm, mode, w, *a, **k)
    cap.update(h=h, W=W, rm=rm, w=w, grad=k.get("grad"), out=out)
    return out
ops.head_loss_rows_stats = hook
orig_or = c_oracle.loss
def hook2(full, lab, mode, **k):
    so, go = orig_or(full, lab, mode, **k)
    cap.update(full=full, lab=lab, go=go, so=so)
    return so, go
c_oracle.loss = hook2
print(S.one_head_case(int(sys.argv[1])))
gs = cap["grad"].float().cpu().numpy()
lab = cap["lab"]
bk = np.argwhere(lab[:, 1:] != -100)
gor = cap["go"][bk[:, 0], bk[:, 1] + 256][:, 31744:32000]
z = cap["full"][bk[:, 0], bk[:, 1] + 256][:, 31744:32000].astype(np.float64)
for r in range(len(bk)):
    p = np.exp(z[r] - z[r].max()); p /= p.sum()
    E = (p * np.arange(1, 257)).sum()
    d = np.abs(gs[r] - gor[r]); i = int(d.argmax())
    print(f"row {r} label {lab[bk[r,0], bk[r,1]+1]} maxp {p.max():.6f} argmax {p.argmax()} E {E:.6f} | worst col {i}: kernel {gs[r,i]:.6e} oracle {gor[r,i]:.6e} p_i {p[i]:.3e} (i+1-E) {i+1-E:.5f} | row max|g| {np.abs(gor[r]).max():.4e}")
print("scalars", cap["so"][:8])
