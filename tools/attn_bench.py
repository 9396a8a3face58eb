"""Micro-benchmark of F.scaled_dot_product_attention fwd / fwd+bwd at the three attention shapes of the bs=64 OpenVLA-7B step."""
import sys, torch, torch.nn.functional as F
dev = "cuda"
if len(sys.argv) > 1:
    print("preferred_rocm_fa_library ->", torch.backends.cuda.preferred_rocm_fa_library(sys.argv[1]))
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, (B, H, T, hd, causal, n, packed) in {"llm": (64, 32, 300, 128, True, 32, False), "dino": (64, 16, 261, 64, False, 23, True), "siglip": (64, 16, 256, 72, False, 26, True),
                                       "siglip_pad96": (64, 16, 256, 96, False, 26, True), "siglip_pad128": (64, 16, 256, 128, False, 26, True)}.items():
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
        leaves = [qkv]
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
    else:
        leaves = [torch.randn(B, T, H, hd, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3)]
        q, k, v = [x.transpose(1, 2) for x in leaves]
    scale = 72 ** -0.5 if name.startswith("siglip") else None
    fwd = t(lambda: F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), is_causal=causal, scale=scale))
    go = torch.randn(B, T, H, hd, device=dev, dtype=torch.bfloat16).transpose(1, 2)
    def fb():
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
        torch.autograd.grad(o, leaves, go)
    both = t(fb)
    io = B * T * H * hd * 2
    print(f"{name}: fwd {fwd:.0f} us, fwd+bwd {both:.0f} us, x{n} layers = {both * n / 1e3:.1f} ms/step; ideal fwd {4 * io / 5e6:.0f} us bwd {8 * io / 5e6:.0f} us", flush=True)
