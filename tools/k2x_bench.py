#!/usr/bin/env python3
"""K2' tile GEMM: the cross-image contraction (embed_dgrad_rows_kernel, VAA_K2E_ROWS=1) against the per-image kernel (VAA_K2E_ROWS=0) —
bitwise equality of K2''s partial tiles and per-dispatch times (vaa_prof_*).   python tools/k2x_bench.py [B ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roboticattack_amd import benchmarks, ops, synthetic  # noqa: E402

DEV = "cuda:0"


def main():
    Bs = [int(v) for v in sys.argv[1:]] or [64, 32, 40, 128, 8]
    D0, D1 = 1024, 1152
    g = torch.Generator(device=DEV).manual_seed(1)
    w0 = (torch.randn(D0, 588, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    w1 = (torch.randn(D1, 588, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    wp0, wp1 = ops.pack_embed_weights(w0.t().contiguous()), ops.pack_embed_weights(w1.t().contiguous())
    for B in Bs:
        img = torch.from_numpy(synthetic.synth_images(3, B, "noise")).to(DEV)
        patch = torch.rand(3, 50, 50, device=DEV, generator=g)
        xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
        xy, th = torch.from_numpy(xy_n).to(DEV), torch.from_numpy(th_n).to(DEV)
        _, _, keep_t, flags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
        dy0 = (torch.randn(B, 256, D0, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
        dy1 = (torch.randn(B, 256, D1, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
        rows = int((flags != 0).sum())
        os.environ["VAA_K2E_ROWS"] = "1"
        ops.prof_start(8)
        for _ in range(4):
            trows = ops.tile_rows_build(flags)
        t_rows = float(np.mean([us for _, us in ops.prof_collect()][1:]))
        assert int(trows[0]) == rows and torch.equal(trows[4 : 4 + rows], (flags.view(-1) != 0).nonzero().view(-1).int())
        res = {}
        for mode in ("0", "1"):
            os.environ["VAA_K2E_ROWS"] = mode
            tr = trows if mode == "1" else None
            for _ in range(3):
                parts = ops.patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, th, keep_t, flags, True, defer_reduce=True, tile_rows=tr)
            torch.cuda.synchronize()
            ops.prof_start(64)
            for _ in range(10):
                parts = ops.patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, th, keep_t, flags, True, defer_reduce=True, tile_rows=tr)
            recs = ops.prof_collect()
            per = {}
            for n, us in recs:
                per.setdefault(n.split("<")[0].strip("("), []).append(us)
            res[mode] = (parts.clone(), {k: float(np.mean(v)) for k, v in per.items()})
        same = torch.equal(res["0"][0], res["1"][0])
        print(f"B={B:4d} rows={rows:5d} ({rows / B:.1f}/img) bitwise_equal={same}  tile_rows_build {t_rows:.2f} us (in the forward, behind K1)")
        for mode in ("0", "1"):
            print("   ", "per-image  " if mode == "0" else "cross-image", {k: round(v, 2) for k, v in res[mode][1].items()}, "sum", round(sum(res[mode][1].values()), 2))
    os.environ.pop("VAA_K2E_ROWS", None)


if __name__ == "__main__":
    main()
