"""Could an unfused attention (batched GEMMs + one softmax pass) beat the flash kernels at T=300? Times the five batched GEMMs."""
import torch
dev = "cuda"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for T in (300, 304, 320):
    N, hd = 2048, 128
    q = torch.randn(N, T, hd, device=dev, dtype=torch.bfloat16)
    k = torch.randn(N, T, hd, device=dev, dtype=torch.bfloat16)
    v = torch.randn(N, T, hd, device=dev, dtype=torch.bfloat16)
    p = torch.randn(N, T, T, device=dev, dtype=torch.bfloat16)
    s_out = torch.empty(N, T, T, device=dev, dtype=torch.bfloat16)
    o_out = torch.empty(N, T, hd, device=dev, dtype=torch.bfloat16)
    r = {"QK^T": t(lambda: torch.bmm(q, k.transpose(1, 2), out=s_out)), "PV": t(lambda: torch.bmm(p, v, out=o_out)),
         "P^T dO": t(lambda: torch.bmm(p.transpose(1, 2), v, out=o_out)), "softmax": t(lambda: torch.softmax(p, -1))}
    print(T, {k_: round(v_) for k_, v_ in r.items()}, flush=True)
