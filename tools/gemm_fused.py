"""Do fused QKV / gate-up projections pay once their shapes are tuned too? (TunableOp tuning on, rotating buffers)"""
import torch, torch.nn.functional as F
dev = "cuda"
M = 19200
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
x = torch.randn(M, 4096, device=dev, dtype=torch.bfloat16)
xs = [torch.randn(M, 4096, device=dev, dtype=torch.bfloat16) for _ in range(4)]  # rotate inputs
for name, N, k in (("qkv", 4096, 3), ("gate_up", 11008, 2)):
    Ws = [torch.randn(N, 4096, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(k)]
    Wc = torch.cat(Ws, 0).contiguous()
    i = [0]
    def sep():
        i[0] = (i[0] + 1) % 4
        return [F.linear(xs[i[0]], w) for w in Ws]
    def fused():
        i[0] = (i[0] + 1) % 4
        return F.linear(xs[i[0]], Wc)
    a, b = t(sep), t(fused)
    print(f"{name}: separate {a:.0f} us, fused {b:.0f} us ({(a - b) / a * 100:+.1f} %)", flush=True)
    # dgrad: dx = sum_i dy_i @ W_i  vs  dy_cat @ W_cat   (TN through transposed weights)
    dys = [torch.randn(M, N, device=dev, dtype=torch.bfloat16) for _ in range(k)]
    dyc = torch.cat(dys, 1).contiguous()
    Wts = [w.t().contiguous() for w in Ws]          # [4096, N]
    Wtc = Wc.t().contiguous()                        # [4096, k*N]
    def dsep():
        dx = F.linear(dys[0], Wts[0])
        for q in range(1, k): dx.addmm_(dys[q], Wts[q].t())
        return dx
    def dfused():
        return F.linear(dyc, Wtc)
    a, b = t(dsep), t(dfused)
    print(f"{name} dgrad: separate {a:.0f} us, fused {b:.0f} us ({(a - b) / a * 100:+.1f} %)", flush=True)
