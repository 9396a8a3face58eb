"""Micro-benchmark: dgrad of a frozen linear as `dy @ W` (NN, what autograd does) vs `F.linear(dy, W^T.contiguous())` (TN)."""
import torch, torch.nn.functional as F, json, sys
from roboticattack_amd.openvla_model import enable_tuned_gemms
enable_tuned_gemms()
dev = "cuda"
M = 64 * 300
out = {}
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (m, o, i) in [(M, 4096, 4096), (M, 11008, 4096), (M, 4096, 11008), (64 * 257, 3072, 1024), (64 * 257, 1024, 4096), (64*256, 4304, 1152), (64*256, 1152, 4304)]:
    W = torch.randn(o, i, device=dev, dtype=torch.bfloat16) * 0.02
    Wt = W.t().contiguous()
    x = torch.randn(m, i, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(m, o, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * m * o * i
    r = {"fwd_us": t(lambda: F.linear(x, W)), "dgrad_nn_us": t(lambda: dy @ W), "dgrad_tn_us": t(lambda: F.linear(dy, Wt))}
    r.update({k.replace("_us", "_PFs"): fl / v / 1e9 for k, v in list(r.items())})
    out[f"{m}x{o}x{i}"] = r
    print(m, o, i, {k: round(v, 3) for k, v in r.items()}, flush=True)
