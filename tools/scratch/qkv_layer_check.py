#!/usr/bin/env python3
"""One Llama layer at OpenVLA-7B dimensions (D=4096, 32 heads, hd=128), bs=8 x 300 tokens: q/k/v as ONE GEMM (VAA_FUSED_QKV=1) against three —
output and input gradient agree to bf16 GEMM summation order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roboticattack_amd.openvla_model import LlamaLayer, openvla_7b_cfg  # noqa: E402

dev = "cuda:0"
cfg = openvla_7b_cfg()
torch.manual_seed(0)
lyr = LlamaLayer(cfg).to(dev).to(torch.bfloat16)
for p in lyr.parameters():
    p.requires_grad_(False)
    if p.dim() == 2:
        p.normal_(0, 0.02)
B, T, D = 8, 300, cfg.llm_dim
hd = D // cfg.llm_heads
x0 = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
go = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
ang = torch.outer(torch.arange(T, device=dev, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd)))
rope_tab = (ang.cos().contiguous(), ang.sin().contiguous())
res = []
for mode in ("0", "1"):
    os.environ["VAA_FUSED_QKV"] = mode
    x = x0.clone().requires_grad_(True)
    y = lyr(x, None, None, rope_tab=rope_tab)
    y.backward(go)
    res.append((y.detach().float(), x.grad.float()))
for name, a, b in (("output", res[0][0], res[1][0]), ("input gradient", res[0][1], res[1][1])):
    rel = float((a - b).abs().max() / b.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))
    print(f"{name}: max |diff| / max |ref| = {rel:.2e}, cosine = {cos:.7f}, finite = {bool(torch.isfinite(b).all())}")
