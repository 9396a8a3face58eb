import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from roboticattack_amd import ops, synthetic
from roboticattack_amd.labels import mask_labels
DEV="cuda:0"; D,V=4096,32064
g = torch.Generator(device=DEV).manual_seed(0)
W = (torch.randn(V, D, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
for R in (128, 16):
    B=R//2
    _, labels, _ = synthetic.synth_text_batch(4242, B)
    labels = mask_labels(labels, [0]).to(DEV)
    rm = ops.LossRowMap(labels)
    h = torch.randn(R, D, device=DEV, generator=g).to(torch.bfloat16)
    cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
    gemms = len(sys.argv) > 1 and sys.argv[1] == "gemms"  # behind ~100 ms of bf16 GEMMs over 3 GB of weights: caches AND clocks as inside a model step
    scratch = (torch.empty(1 << 30, dtype=torch.uint8, device=DEV), torch.empty(1 << 30, dtype=torch.uint8, device=DEV)) if cold else None
    if gemms:
        ws_ = [torch.randn(4096, 11008, device=DEV, dtype=torch.bfloat16) * 0.01 for _ in range(32)]
        xs_ = torch.randn(16384, 4096, device=DEV, dtype=torch.bfloat16)
    for _ in range(5):
        if cold:  # a 1 GiB device copy replaces every L2 and the Infinity Cache: what the call sees inside a model step
            scratch[1].copy_(scratch[0])
        if gemms:
            for w_ in ws_:
                y_ = xs_ @ w_
            hh = h + 0  # the hidden rows are fresh, as the final norm leaves them
            o = ops.head_slice_fwd_bwd(hh, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=False)
            continue
        o = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=False)
    torch.cuda.synchronize()
    del scratch
    nwg = (R+15)//16*(16 if os.environ.get('VAA_K3S_COLS') == '16' else 32)
    st = o["zs"][512<<10:(512<<10)+nwg*16*8].view(torch.int64).view(nwg,16).cpu().numpy().astype(np.int64)
    t0 = st[:,0].min()
    rel = (st[:, :10]-t0)*0.01  # us
    names = ["start","ph1 loop end","zs stored","B issued+bar","first poll ok","stats done","sync","grad done","mfma done","end"]
    print(f"R={R}: per-stamp [min / median / max] us over {nwg} workgroups")
    for i,n in enumerate(names):
        print(f"   {n:16s} {rel[:,i].min():7.2f} {np.median(rel[:,i]):7.2f} {rel[:,i].max():7.2f}")
