#!/usr/bin/env python3
"""LM head fused with K3's statistics (vaa_head_loss_rows_stats, csrc/vaa_head.hip) against the LM-head GEMM (hipBLASLt) + vaa_loss_rows_stats:
agreement, per-dispatch / in-stream / cold-cache times.   python tools/head_bench.py [R ...]
(R = labelled rows: 128 at bs=64 with maskidx=[0], 16 at bs=8)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.labels import mask_labels  # noqa: E402

head_loss_rows_stats = ops.head_loss_rows_stats

DEV = "cuda:0"
D, V = 4096, 32064


def stream_time(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def cold_time(fn, scratch, n=12):
    """one call at a time behind a 1 GiB device copy that replaces every L2 and the 256 MB Infinity Cache: what the call costs inside a model
    step, where 15+ GB of other weights pass between two uses of the head weight"""
    ts = []
    for _ in range(n):
        scratch[1].copy_(scratch[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[2:]))


def main():
    from roboticattack_amd.openvla_model import enable_tuned_gemms

    enable_tuned_gemms()
    g = torch.Generator(device=DEV).manual_seed(0)
    W = (torch.randn(V, D, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    for R in [int(v) for v in sys.argv[1:]] or [128, 16, 64, 32]:
        B = R // 2
        _, labels, _ = synthetic.synth_text_batch(4242, B)
        labels = mask_labels(labels, [0]).to(DEV)
        assert int((labels[:, 1:] != -100).sum()) == R
        rm = ops.LossRowMap(labels)
        h = (torch.randn(R, D, device=DEV, generator=g) * 1.0).to(torch.bfloat16)
        gs_ref = torch.empty((R, 256), dtype=torch.bfloat16, device=DEV)
        gs_new = torch.empty_like(gs_ref)
        msg = torch.zeros(7504, device=DEV)
        parts = torch.zeros((64, 7500), device=DEV)

        def ref():
            logits = torch.nn.functional.linear(h, W)
            return ops.loss_rows_stats(logits, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs_ref)

        def new():
            return head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs_new)

        # agreement: scalars / predictions through the epilogue's fold of either workspace
        sc_ref, sc_new = torch.zeros(8, device=DEV), torch.zeros(8, device=DEV)
        ws = ref()
        p_ref = ops.step_epilogue(parts, msg, sc_ref, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws)
        ws, lg = head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs_new, want_logits=True)
        p_new = ops.step_epilogue(parts, msg, sc_new, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws)
        torch.cuda.synchronize()
        lg_ref = torch.nn.functional.linear(h, W)
        dl = (lg.float() - lg_ref.float()).abs()
        print(f"R={R:4d}: logits vs hipBLASLt bf16: max |d| {float(dl.max()):.4f} (scale {float(lg_ref.float().abs().max()):.2f}), differing {float((dl > 0).float().mean()) * 100:.2f} %")
        print("        scalars ref", [round(float(v), 6) for v in sc_ref], "\n        scalars new", [round(float(v), 6) for v in sc_new])
        print("        pred_full equal:", bool(torch.equal(p_ref[1], p_new[1])), " pred_slice equal:", bool(torch.equal(p_ref[0], p_new[0])),
              " grad slice max |d| / max:", float((gs_new.float() - gs_ref.float()).abs().max() / gs_ref.float().abs().max()))
        # the same statistics kernel fed with the fused kernel's own bf16 logits: slice statistics and gradient must be bitwise
        gs_chk = torch.empty_like(gs_ref)
        sc_chk = torch.zeros(8, device=DEV)
        ws = ops.loss_rows_stats(lg, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs_chk)
        ops.step_epilogue(parts, msg, sc_chk, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws)
        torch.cuda.synchronize()
        print("        on the SAME bf16 logits: grad slice bitwise", bool(torch.equal(gs_chk.view(torch.int16), gs_new.view(torch.int16))),
              " scalars max rel d", float(((sc_chk - sc_new).abs() / sc_chk.abs().clamp_min(1e-9)).max()))
        # times
        t_ref, t_new = stream_time(ref), stream_time(new)
        ops.prof_start(64)
        for _ in range(6):
            new()
        recs = ops.prof_collect()
        per = {}
        for n, us in recs[2:]:
            per.setdefault(n.split("<")[0].strip("("), []).append(us)
        lin = stream_time(lambda: torch.nn.functional.linear(h, W))
        scratch = (torch.empty(1 << 30, dtype=torch.uint8, device=DEV), torch.empty(1 << 30, dtype=torch.uint8, device=DEV))
        c_ref, c_new, c_lin = cold_time(ref, scratch), cold_time(new, scratch), cold_time(lambda: torch.nn.functional.linear(h, W), scratch)
        del scratch
        print(f"        COLD (behind a 1 GiB copy, events around one call): GEMM + K3 stats {c_ref:.1f} us (GEMM alone {c_lin:.1f}) | fused {c_new:.1f} us")
        print(f"        in a stream: GEMM + K3 stats {t_ref:.1f} us (GEMM alone {lin:.1f}) | fused {t_new:.1f} us  per dispatch {({k: round(float(np.mean(v)), 1) for k, v in per.items()})}"
              f"  -> weight stream {V * D * 2 / 1e6 / (per.get('head_stats_kernel', [1e9])[0]):.0f} GB/s... ", end="")
        hs = float(np.mean(per.get("head_stats_kernel", [float('nan')])))
        print(f"{V * D * 2 / hs / 1e3:.0f} GB/s = {V * D * 2 / hs / 1e3 / 8000:.2f} of 8 TB/s")


if __name__ == "__main__":
    main()
