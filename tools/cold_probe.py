#!/usr/bin/env python3
"""Where the in-step penalty of the hot-path kernels comes from: the five launches of the step timed per dispatch (vaa_prof_*) (a) back to back
with warm caches, (b) each after a 1 GiB device copy that replaces the contents of every L2 and of the Infinity Cache (cold inputs, idle
clocks), (c) each after ~2 ms of bf16 GEMMs (cold inputs AND the clocks of a loaded part — what the attack step looks like).
  python tools/cold_probe.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roboticattack_amd import ops, synthetic  # noqa: E402
from roboticattack_amd.benchmarks import random_params  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
img = torch.from_numpy(synthetic.synth_images(1234, min(B, 64), "noise")).to(dev)
patch = torch.rand(3, 50, 50, device=dev)
xy_n, th_n = random_params(B, 50, 50, 42)
xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
dy0 = (torch.randn(B, 256, 1024, device=dev, generator=g) * 0.1).to(torch.bfloat16)
dy1 = (torch.randn(B, 256, 1152, device=dev, generator=g) * 0.1).to(torch.bfloat16)
wp0 = ops.pack_embed_weights((torch.randn(588, 1024, device=dev, generator=g) * 0.05).to(torch.bfloat16))
wp1 = ops.pack_embed_weights((torch.randn(588, 1152, device=dev, generator=g) * 0.05).to(torch.bfloat16))
big_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
big_b = torch.empty_like(big_a)
ga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
gb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def flush():
    big_b.copy_(big_a)


def load():
    for _ in range(4):
        torch.mm(ga, gb)
    big_b.copy_(big_a)


def one(between):
    _, _, keep_t, flags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
    between()
    ops.patch_embed_grad_gather_tiles(dy0, dy1, wp0, wp1, patch, xy, th, keep_t, flags, True, defer_reduce=True)


for tag, between in (("warm, back to back", lambda: None), ("after a 1 GiB copy", flush), ("after 4 GEMMs + the copy", load)):
    for _ in range(3):
        one(between)
    torch.cuda.synchronize()
    ops.prof_start(1024)
    for _ in range(20):
        between()
        one(between)
    torch.cuda.synchronize()
    per = {}
    for n, us in ops.prof_collect():
        per.setdefault(n.lstrip("(").replace("vaa::", "").split("<")[0].split("(")[0], []).append(us)
    print(f"B={B} {tag}: " + ", ".join(f"{k} {np.mean(v):.2f}" for k, v in per.items()))
