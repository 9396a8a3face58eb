"""Correctness + timing of the hand-written attention against F.scaled_dot_product_attention (GPU)."""
import sys, torch, torch.nn.functional as F
from roboticattack_amd import model_ops
dev = "cuda"
torch.manual_seed(0)
def ref(q, k, v, causal, scale):
    o = F.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), is_causal=causal, scale=scale)
    return o.transpose(1, 2)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
ok = True
for (B, H, T, hd, causal, packed) in [(2, 4, 300, 128, True, False), (2, 3, 261, 64, False, True), (2, 3, 256, 72, False, True), (1, 2, 17, 128, True, False),
                                      (3, 2, 64, 64, True, True), (1, 9, 65, 72, False, True), (2, 2, 130, 128, False, False), (1, 1, 1, 64, True, False), (1, 2, 333, 128, True, False)]:
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device=dev).to(torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = [torch.randn(B, T, H, hd, device=dev).to(torch.bfloat16) * 1.5 for _ in range(3)]
    scale = hd ** -0.5
    o, lse = model_ops.attention_fwd(q, k, v, causal, scale)
    r = ref(q, k, v, causal, scale)
    s = torch.einsum("bthd,bshd->bhts", q.float(), k.float()) * scale
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=dev, dtype=torch.bool).triu(1), float("-inf"))
    lr = torch.logsumexp(s, -1)
    e, el = (o.float() - r).abs().max().item(), (lse - lr).abs().max().item()
    good = e < 2e-2 and el < 2e-3
    ok &= good
    # backward against fp32 autograd of the reference formulation
    go = torch.randn(B, T, H, hd, device=dev).to(torch.bfloat16)
    qf, kf, vf = [x.detach().float().requires_grad_(True) for x in (q, k, v)]
    ref(qf, kf, vf, causal, scale).backward(go.float())
    dq, dk, dv = model_ops.attention_bwd(q, k, v, o, lse, go, causal, scale)
    eb = [((d.float() - r_.grad).abs().max() / (r_.grad.abs().max() + 1e-2)).item() for d, r_ in ((dq, qf), (dk, kf), (dv, vf))]
    good &= max(eb) < 2e-2
    ok &= good
    print(f"B{B} H{H} T{T} hd{hd} causal={causal} packed={packed}: max|o-ref|={e:.2e} max|lse-ref|={el:.2e} rel dq/dk/dv err {eb[0]:.1e} {eb[1]:.1e} {eb[2]:.1e} {'ok' if good else 'FAIL'}", flush=True)
for name, (B, H, T, hd, causal, packed) in {"llm": (64, 32, 300, 128, True, False), "dino": (64, 16, 261, 64, False, True), "siglip": (64, 16, 256, 72, False, True)}.items():
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device=dev).to(torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = [torch.randn(B, T, H, hd, device=dev).to(torch.bfloat16) for _ in range(3)]
    mine = t(lambda: model_ops.attention_fwd(q, k, v, causal, hd ** -0.5))
    qq, kk, vv = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    sd = t(lambda: F.scaled_dot_product_attention(qq, kk, vv, is_causal=causal, scale=hd ** -0.5))
    o, lse = model_ops.attention_fwd(q, k, v, causal, hd ** -0.5)
    go = torch.randn(B, T, H, hd, device=dev).to(torch.bfloat16)
    mineb = t(lambda: model_ops.attention_bwd(q, k, v, o, lse, go, causal, hd ** -0.5, packed_grad=packed))
    leaves = [x.detach().requires_grad_(True) for x in ((qkv,) if packed else (q, k, v))]
    if packed:
        qq, kk, vv = leaves[0].permute(2, 0, 3, 1, 4)
    else:
        qq, kk, vv = [x.transpose(1, 2) for x in leaves]
    got = go.transpose(1, 2)
    def fb():
        oo = F.scaled_dot_product_attention(qq, kk, vv, is_causal=causal, scale=hd ** -0.5)
        torch.autograd.grad(oo, leaves, got)
    sdb = t(fb)
    print(f"{name}: hand-written fwd {mine:.0f} us bwd {mineb:.0f} us | SDPA fwd {sd:.0f} us fwd+bwd {sdb:.0f} us", flush=True)
print("ALL OK" if ok else "SOME FAILED")
