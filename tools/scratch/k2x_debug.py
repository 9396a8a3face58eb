import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from roboticattack_amd import benchmarks, ops, synthetic, _lib
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D0, D1 = 1024, 1152
g = torch.Generator(device=DEV).manual_seed(1)
w0 = (torch.randn(D0, 588, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
w1 = (torch.randn(D1, 588, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
wp0, wp1 = ops.pack_embed_weights(w0.t().contiguous()), ops.pack_embed_weights(w1.t().contiguous())
img = torch.from_numpy(synthetic.synth_images(3, B, "noise")).to(DEV)
patch = torch.rand(3, 50, 50, device=DEV, generator=g)
xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(DEV), torch.from_numpy(th_n).to(DEV)
_, _, keep_t, flags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
dy0 = (torch.randn(B, 256, D0, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
dy1 = (torch.randn(B, 256, D1, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
os.environ["VAA_K2E_ROWS"] = "1"
trows = ops.tile_rows_build(flags)
L = _lib.lib()
ws = ops._workspace(patch.device, L.vaa_patch_embed_grad_ws_bytes(B, 50, 50), "k2e")
ws.zero_()
parts = ops.patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, th, keep_t, flags, True, defer_reduce=True, tile_rows=trows)
torch.cuda.synchronize()
part_bytes = (L.vaa_patch_grad_ws_bytes(B, 50, 50) + 255) // 256 * 256
planes = ws[part_bytes : part_bytes + 2 * B * 256 * 588 * 2].view(torch.bfloat16).view(2, B * 256, 588)
idx = (flags.view(-1) != 0).nonzero().view(-1)
for t, (dy, w) in enumerate(((dy0, w0), (dy1, w1))):
    ref = (dy.view(B * 256, -1)[idx].float() @ w.float()).to(torch.bfloat16)
    got = planes[t][idx]
    bad = (got.float() - ref.float()).abs() > 0.02 * ref.float().abs().max()
    print("tower", t, "rows", idx.numel(), "bad elements", int(bad.sum()), "of", bad.numel(), "bad rows", int(bad.any(1).sum()), "bad cols", int(bad.any(0).sum()))
    if bad.any():
        br = bad.any(1).nonzero().view(-1)
        print("  first bad rows (compact idx):", br[:20].tolist(), " groups:", sorted(set((br // 64).tolist()))[:20])
        bc = bad.any(0).nonzero().view(-1)
        print("  bad col blocks:", sorted(set((bc // 16).tolist())))
        r0 = int(br[0]); print("  got", got[r0, :8].float().tolist(), "ref", ref[r0, :8].float().tolist())
# which partial K-sum does `got` match?
dy, w, t = dy0, w0, 0
A = dy.view(B * 256, -1)[idx].float()
got = planes[t][idx].float()
for lo, hi in ((0, 384), (384, 768), (768, 1024), (0, 768), (0, 1024), (0, 192), (192, 384)):
    ref = A[:, lo:hi] @ w.float()[lo:hi]
    err = (got - ref).abs().max() / ref.abs().max()
    print("k in", (lo, hi), "rel err", float(err))
