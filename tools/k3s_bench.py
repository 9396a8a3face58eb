#!/usr/bin/env python3
"""K3s (vaa_head_slice_fwd_bwd, csrc/vaa_head_slice.hip) against what it replaces — K3h (vaa_head_loss_rows_stats + finish) followed by the 256-column
head-backward GEMM: per-dispatch (the library's own events), in-stream and cold-cache times.   python tools/k3s_bench.py [R ...]
(R = labelled rows: 128 at bs=64 with maskidx=[0], 16 at bs=8)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.labels import mask_labels  # noqa: E402
from tools.head_bench import cold_time, stream_time  # noqa: E402

DEV = "cuda:0"
D, V = 4096, 32064


def per_dispatch(fn, n=12):
    ops.prof_start(256)
    for _ in range(n):
        fn()
    recs = ops.prof_collect()
    per = {}
    for name, us in recs:
        per.setdefault(name.split("<")[0].strip("("), []).append(us)
    return {k: round(float(np.median(v[len(v) // 3:])), 2) for k, v in per.items()}


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    W = (torch.randn(V, D, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    for R in [int(v) for v in sys.argv[1:]] or [128, 16, 64, 32]:
        B = R // 2
        _, labels, _ = synthetic.synth_text_batch(4242, B)
        labels = mask_labels(labels, [0]).to(DEV)
        assert int((labels[:, 1:] != -100).sum()) == R
        rm = ops.LossRowMap(labels)
        h = torch.randn(R, D, device=DEV, generator=g).to(torch.bfloat16)
        gs = torch.empty((R, 256), dtype=torch.bfloat16, device=DEV)

        def old():
            ops.head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs)
            return gs @ W[31744:32000]

        def new():
            return ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=False)["dh"]

        def new_pub():
            return ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=True)["dh"]

        a, b = old().double(), new().double()
        torch.cuda.synchronize()
        print(f"R={R:4d}: dH K3s vs K3h + GEMM: max |d| {float((a - b).abs().max()):.3e} (scale {float(a.abs().max()):.3e})")
        t_old, t_new, t_pub = stream_time(old), stream_time(new), stream_time(new_pub)
        os.environ["VAA_K3S_ONE_LAUNCH"] = "0"
        t_two = stream_time(new)
        p_two = per_dispatch(new)
        del os.environ["VAA_K3S_ONE_LAUNCH"]
        p_old, p_new, p_pub = per_dispatch(old), per_dispatch(new), per_dispatch(new_pub)
        os.environ["VAA_K3S_DEBUG_PHASES"] = "1"
        p_ph1 = per_dispatch(new)
        os.environ["VAA_K3S_DEBUG_PHASES"] = "2"
        p_ph2 = per_dispatch(new)
        del os.environ["VAA_K3S_DEBUG_PHASES"]
        print(f"        phases alone: logits {p_ph1} | statistics + gradient + dH {p_ph2}")
        scratch = (torch.empty(1 << 30, dtype=torch.uint8, device=DEV), torch.empty(1 << 30, dtype=torch.uint8, device=DEV))
        c_old, c_new = cold_time(old, scratch), cold_time(new, scratch)
        del scratch
        print(f"        in a stream: K3h + finish + GEMM {t_old:.1f} us | K3s {t_new:.1f} us (publishing scalars {t_pub:.1f}; two launches {t_two:.1f})")
        print(f"        COLD (behind a 1 GiB copy): K3h + finish + GEMM {c_old:.1f} us | K3s {c_new:.1f} us")
        print(f"        per dispatch: old {p_old} | K3s {p_new} | K3s publishing {p_pub} | K3s two launches {p_two}")
        byt = R * D * 2 * 2 + 2 * 256 * D * 2
        k = p_new.get("head_slice_kernel", float("nan"))
        print(f"        K3s algorithmic bytes {byt / 1e6:.2f} MB / {k} us = {byt / k / 1e6:.3f} TB/s")


if __name__ == "__main__":
    main()
