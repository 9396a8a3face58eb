#!/usr/bin/env python3
"""Do independent per-rank-shaped GEMMs of one Llama layer overlap when they are issued on separate HIP streams?  (bs=8 per rank: M = 2,400 rows)
q/k/v: three [M,4096] x [4096,4096]; gate/up: two [M,4096] x [4096,11008]. Sequential on one stream against one stream per GEMM (fork / join by events),
with the shipped hipBLASLt selections; and, for reference, the same product as ONE GEMM over the row-concatenated weights (library default selection).  python tools/gemm_overlap_probe.py [M]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import openvla_model  # noqa: E402

openvla_model.enable_tuned_gemms()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
x = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
for name, n, cnt in (("q/k/v", 4096, 3), ("gate/up", 11008, 2)):
    ws = [torch.randn(n, 4096, device=dev, dtype=torch.bfloat16) for _ in range(cnt)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(cnt - 1)]

    def seq():
        return [torch.nn.functional.linear(x, w) for w in ws]

    def par():
        cur = torch.cuda.current_stream()
        outs = [None] * cnt
        for s in streams:
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i + 1] = torch.nn.functional.linear(x, ws[i + 1])
        outs[0] = torch.nn.functional.linear(x, ws[0])
        for s in streams:
            cur.wait_stream(s)
        return outs

    wcat = torch.cat(ws, 0)

    def fused():
        return torch.nn.functional.linear(x, wcat)

    res = {}
    for tag, fn in (("one stream", seq), ("one stream per GEMM", par), ("ONE GEMM over the concatenated weights (untuned selection)", fused)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * M * 4096 * n * cnt
    print(f"M={M} {name}: " + ", ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TFLOP/s)" for k, v in res.items()))
