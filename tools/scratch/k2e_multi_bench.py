"""K2' in per-image mode (config 5: 100x100 base, sizes 61..139) against the unfused sequence (two dgrad GEMMs, fold, cat, K2 multi)."""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from roboticattack_amd import benchmarks, ops, synthetic
from roboticattack_amd.transform import RandomPatchTransform
dev = torch.device("cuda:0")
D0, D1 = 1024, 1152
for B in (4, 32):
    random.seed(3); np.random.seed(3)
    sizes, xy_n, th_n = RandomPatchTransform("cpu", True)._draw_resized(B, 100, 100, True)
    pdesc_n, total = ops.make_pdesc(sizes)
    pdesc, xy, th = torch.from_numpy(pdesc_n).to(dev), torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    mh = (int(pdesc_n[:, 0].max()), int(pdesc_n[:, 1].max()))
    img = torch.from_numpy(synthetic.synth_images(1, B, "noise")).to(dev)
    packed = ops.patch_resize_fwd(torch.rand(3, 100, 100, device=dev), pdesc, total)
    _, keep = ops.patch_apply_fwd_multi(img, packed, pdesc, mh, xy, th, True)
    dy0 = (torch.randn(B, 256, D0, device=dev) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, D1, device=dev) * 0.1).to(torch.bfloat16)
    w0 = (torch.randn(D0, 588, device=dev) * 0.05).to(torch.bfloat16)
    w1 = (torch.randn(D1, 588, device=dev) * 0.05).to(torch.bfloat16)
    wp0, wp1 = ops.pack_embed_weights(w0.t().contiguous()), ops.pack_embed_weights(w1.t().contiguous())
    def unfused():
        fold = lambda dy, w: (dy @ w).view(B, 16, 16, 3, 14, 14).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 224, 224)
        return ops.patch_grad_gather_multi(torch.cat([fold(dy0, w0), fold(dy1, w1)], dim=1), packed, pdesc, mh, xy, th, keep, True)
    fused = lambda: ops.patch_embed_grad_gather_multi(dy0, dy1, wp0, wp1, packed, pdesc, mh, xy, th, keep, True)
    print("B", B, "unfused %.1f us   K2' multi %.1f us" % (benchmarks._time(unfused, 20)[0] * 1e6, benchmarks._time(fused, 20)[0] * 1e6))
