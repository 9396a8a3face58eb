#!/usr/bin/env python3
"""Model-side attention kernels (csrc/vaa_attention.hip) alone, at the shapes of the bs=64 OpenVLA step: times per call with the operands
cycled over several buffer sets (cold L2 / MALL, like inside the step) and a numerics check against fp32 softmax attention.

    python tools/attn_bench.py [--shape llm|dino|siglip|all] [--bs 64] [--iters 20] [--check]
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o a -- python tools/attn_bench.py --shape llm      (per-kernel split)

Shapes: llm = Llama layer (T = 300, 32 heads x 128, causal, rotary adjoint in the backward epilogues, q/k/v separate [B,T,H,hd]);
dino = DINOv2-L block (T = 261, 16 x 64, packed qkv); siglip = SigLIP so400m block (T = 256, 16 x 72, packed qkv).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = {"llm": (300, 32, 128, True, False), "dino": (261, 16, 64, False, True), "siglip": (256, 16, 72, False, True)}


def make(B, T, H, hd, packed, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device=dev, generator=g).to(torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = [torch.randn(B, T, H, hd, device=dev, generator=g).to(torch.bfloat16) for _ in range(3)]
    go = torch.randn(B, T, H, hd, device=dev, generator=g).to(torch.bfloat16)
    return q, k, v, go


def rope_tab(T, hd, dev):
    ang = torch.outer(torch.arange(T, device=dev, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd)))
    return ang.cos().to(torch.bfloat16).float().contiguous(), ang.sin().to(torch.bfloat16).float().contiguous()


def check(name, T, H, hd, causal, packed, dev):
    from roboticattack_amd import model_ops

    B = 2
    q, k, v, go = make(B, T, H, hd, packed, dev, 7)
    scale = hd ** -0.5
    qf, kf, vf = [x.detach().float().requires_grad_(True) for x in (q, k, v)]
    s = torch.einsum("bthd,bshd->bhts", qf, kf) * scale
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=dev, dtype=torch.bool).triu(1), float("-inf"))
    ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vf)
    ref.backward(go.float())
    o, lse = model_ops.attention_fwd(q, k, v, causal, scale)
    grads = model_ops.attention_bwd(q, k, v, o, lse, go, causal, scale, packed_grad=packed)
    if packed:
        grads = (grads[:, :, 0], grads[:, :, 1], grads[:, :, 2])
    errs = {"o": float((o.float() - ref).abs().max() / ref.abs().max()), "lse": float((lse - torch.logsumexp(s, -1)).abs().max())}
    for n, d, r in zip(("dq", "dk", "dv"), grads, (qf.grad, kf.grad, vf.grad)):
        errs[n] = float((d.float() - r).abs().max() / r.abs().max())
    ok = errs["o"] <= 2 ** -7 + 1e-3 and errs["lse"] < 1e-4 and all(errs[n] <= 2e-2 for n in ("dq", "dk", "dv"))
    print(json.dumps({"check": name, "ok": ok, **{k_: round(v_, 6) for k_, v_ in errs.items()}}), flush=True)
    return ok


def bench(name, B, T, H, hd, causal, packed, iters, dev, nsets=4):
    from roboticattack_amd import model_ops

    sets = [make(B, T, H, hd, packed, dev, 11 + i) for i in range(nsets)]
    scale = hd ** -0.5
    rope = rope_tab(T, hd, dev) if name == "llm" else None
    outs = [model_ops.attention_fwd(q, k, v, causal, scale) for q, k, v, _ in sets]
    torch.cuda.synchronize()

    def timed(fn):
        for i in range(3):
            fn(i % nsets)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i % nsets)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    t_f = timed(lambda i: model_ops.attention_fwd(sets[i][0], sets[i][1], sets[i][2], causal, scale))
    t_b = timed(lambda i: model_ops.attention_bwd(sets[i][0], sets[i][1], sets[i][2], outs[i][0], outs[i][1], sets[i][3], causal, scale,
                                                  packed_grad=packed, rope=rope))
    esz = 2 * B * T * H * hd
    rec = {"shape": name, "B": B, "T": T, "H": H, "hd": hd, "causal": causal, "fwd_us": round(t_f, 1), "bwd_us": round(t_b, 1),
           "fwd_floor_us_at_6.3TBs": round(4 * esz / 6.3e6, 1), "bwd_two_kernel_floor_us": round(12 * esz / 6.3e6, 1), "tensor_MB": round(esz / 1e6, 1)}
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="all")
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    names = list(SHAPES) if a.shape == "all" else [a.shape]
    ok = True
    for n in names:
        T, H, hd, causal, packed = SHAPES[n]
        if a.check:
            ok &= check(n, T, H, hd, causal, packed, dev)
        bench(n, a.bs, T, H, hd, causal, packed, a.iters, dev)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
