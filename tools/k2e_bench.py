"""K2' (patch-embed backward on the kept tiles + gather) against the unfused sequence it replaces: two dgrad GEMMs, fold, K2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import benchmarks, ops, synthetic
dev = torch.device("cuda:0")
B, D0, D1 = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 1024, 1152
img = torch.from_numpy(synthetic.synth_images(1, B, "noise")).to(dev)
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
_, keep = ops.patch_apply_fwd(img, patch, xy, th, True, want_keep=True)
dy0 = (torch.randn(B, 256, D0, device=dev) * 0.1).to(torch.bfloat16)
dy1 = (torch.randn(B, 256, D1, device=dev) * 0.1).to(torch.bfloat16)
w0 = (torch.randn(D0, 588, device=dev) * 0.05).to(torch.bfloat16)
w1 = (torch.randn(D1, 588, device=dev) * 0.05).to(torch.bfloat16)
wt0, wt1 = ops.pack_embed_weights(w0.t().contiguous()), ops.pack_embed_weights(w1.t().contiguous())
def unfused():
    def fold(dy, w):
        return (dy @ w).view(B, 16, 16, 3, 14, 14).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 224, 224)
    gout = torch.cat([fold(dy0, w0), fold(dy1, w1)], dim=1)
    return ops.patch_grad_gather(gout, patch, xy, th, keep, True)
def fused():
    return ops.patch_embed_grad_gather(dy0, dy1, wt0, wt1, patch, xy, th, keep, True)
print("B", B, "unfused (2 GEMMs + fold + cat + K2): %.1f us" % (benchmarks._time(unfused, 30)[0] * 1e6))
print("K2' fused: %.1f us" % (benchmarks._time(fused, 30)[0] * 1e6))
