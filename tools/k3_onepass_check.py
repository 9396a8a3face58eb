#!/usr/bin/env python3
"""K3 full-row gradient (UADA / CE): the experimental one-pass form (VAA_K3_ONE_PASS=1: grid-wide hand-over, logits read once) against the two-launch product form — bitwise equality of every
output and per-dispatch times (vaa_prof_*).  python tools/k3_onepass_check.py [R-defining batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.labels import mask_labels  # noqa: E402

dev = torch.device("cuda:0")
for B, maskidx, mode, dtype in ((64, [0], ops.LOSS_UADA, torch.bfloat16), (16, [0], ops.LOSS_UADA, torch.bfloat16), (8, list(range(7)), ops.LOSS_CE, torch.bfloat16),
                                (64, [0, 1, 2], ops.LOSS_UADA, torch.float32)):
    _, labels, _ = synthetic.synth_text_batch(4242, B)
    labels = mask_labels(labels, maskidx).to(dev)
    R = int((labels[:, 1:] != -100).sum())
    logits = (torch.randn(R, 32064, device=dev) * 2).to(dtype)
    rm = ops.LossRowMap(labels)
    res, times, wall = {}, {}, {}
    for tag, env in (("two_launch", "0"), ("one_pass", "1")):
        os.environ["VAA_K3_ONE_PASS"] = env
        g = torch.empty_like(logits)
        for _ in range(3):
            out = ops.loss_rows_fwd_bwd(logits, rm, mode, w=5.0, grad_kind=ops.GRAD_FULL, grad=g)
        torch.cuda.synchronize()
        ops.prof_start(256)
        for _ in range(30):
            out = ops.loss_rows_fwd_bwd(logits, rm, mode, w=5.0, grad_kind=ops.GRAD_FULL, grad=g)
        torch.cuda.synchronize()
        recs = ops.prof_collect()
        per = {}
        for n, us in recs:
            per.setdefault(n, []).append(us)
        times[tag] = {n: float(np.mean(v)) for n, v in per.items()}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            out = ops.loss_rows_fwd_bwd(logits, rm, mode, w=5.0, grad_kind=ops.GRAD_FULL, grad=g)
        e1.record()
        torch.cuda.synchronize()
        wall[tag] = e0.elapsed_time(e1) * 1e3 / 200
        res[tag] = (out[0].clone(), out[1].clone(), out[2].clone(), g.clone())
    same = all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
               for a, b in zip(res["two_launch"], res["one_pass"]))
    t2, t1 = sum(times["two_launch"].values()), sum(times["one_pass"].values())
    print(f"B={B} R={R} mode={mode} {dtype}: bitwise equal {same}; two launches {t2:.2f} us {[round(v, 2) for v in times['two_launch'].values()]}, one pass {t1:.2f} us; in-stream per call (200 calls between two events): {wall['two_launch']:.2f} vs {wall['one_pass']:.2f} us")
