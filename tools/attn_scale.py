"""Forward/backward time of the attention kernels vs sequence length (per-tile cost vs per-workgroup overhead)."""
import sys, torch
from roboticattack_amd import model_ops
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for causal in (True, False):
    for (B, T) in [(64, 300), (64, 320), (32, 640), (16, 1280), (8, 2560)]:
        H, hd = 32, 128
        q, k, v, go = [torch.randn(B, T, H, hd, device="cuda").to(torch.bfloat16) for _ in range(4)]
        f = t(lambda: model_ops.attention_fwd(q, k, v, causal, hd ** -0.5))
        o, lse = model_ops.attention_fwd(q, k, v, causal, hd ** -0.5)
        bw = t(lambda: model_ops.attention_bwd(q, k, v, o, lse, go, causal, hd ** -0.5))
        nt = (T + 63) // 64
        iters = B * H * (nt * (nt + 1) // 2 if causal else nt * nt)
        fl = 4.0 * B * H * T * T * hd * (0.5 if causal else 1.0)
        print(f"causal={causal} B={B} T={T}: fwd {f:.0f} us ({fl / f / 1e6:.0f} TF/s, {f * 1e3 * 256 / iters:.0f} ns/tile-iter/CU) bwd {bw:.0f} us ({2.5 * fl / bw / 1e6:.0f} TF/s)", flush=True)
