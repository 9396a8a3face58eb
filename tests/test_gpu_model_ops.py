"""Optional model-side fused operators (include/vaa_model_ops.h) against the eager PyTorch formulation they replace."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rope_ref(q, k, cos, sin):
    from roboticattack_amd.openvla_model import _rope

    return _rope(q, k, cos, sin)


def test_rope_fwd_bwd_matches_eager():
    from roboticattack_amd import model_ops

    torch.manual_seed(0)
    B, T, H, hd = 3, 37, 4, 128
    x = torch.randn(B, T, H * hd, device=DEV, dtype=torch.bfloat16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd))
    ang = torch.outer(torch.arange(T, device=DEV, dtype=torch.float32), inv)
    full = torch.cat([ang, ang], -1)
    cos, sin = full.cos().to(torch.bfloat16)[None, None], full.sin().to(torch.bfloat16)[None, None]
    tab = (ang.cos().to(torch.bfloat16).float().contiguous(), ang.sin().to(torch.bfloat16).float().contiguous())
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    qa = xa.view(B, T, H, hd).transpose(1, 2)
    ref, _ = _rope_ref(qa.float(), qa.float(), cos.float(), sin.float())  # fp32 reference of the same formula
    got = model_ops.RopeFn.apply(xb.view(B, T, H, hd), *tab).transpose(1, 2)
    assert got.dtype == torch.bfloat16 and got.shape == ref.shape
    assert (got.float() - ref).abs().max() <= 2 ** -7 * ref.abs().max()  # one bf16 rounding
    g = torch.randn_like(ref)
    ref.backward(g)
    got.backward(g.to(torch.bfloat16))
    assert (xb.grad.float() - xa.grad.float()).abs().max() <= 2 ** -6 * xa.grad.float().abs().max()
    # strided input (e.g. a gradient arriving in [B,H,T,hd] memory order) takes the same kernel
    y = torch.randn(B, H, T, hd, device=DEV, dtype=torch.bfloat16)
    s1 = model_ops.RopeFn.apply(y.transpose(1, 2), *tab)
    s2 = model_ops.RopeFn.apply(y.transpose(1, 2).contiguous(), *tab)
    assert torch.equal(s1, s2)


def test_swiglu_fwd_bwd_matches_eager():
    import torch.nn.functional as F

    from roboticattack_amd import model_ops

    torch.manual_seed(1)
    g = (torch.randn(50, 11008, device=DEV) * 2).to(torch.bfloat16)
    u = torch.randn(50, 11008, device=DEV).to(torch.bfloat16)
    ga, ua = g.float().requires_grad_(True), u.float().requires_grad_(True)
    gb, ub = g.clone().requires_grad_(True), u.clone().requires_grad_(True)
    ref = F.silu(ga) * ua
    got = model_ops.SwiGLUFn.apply(gb, ub)
    assert (got.float() - ref).abs().max() <= 2 ** -7 * ref.abs().max()
    dy = torch.randn_like(ref)
    ref.backward(dy)
    got.backward(dy.to(torch.bfloat16))
    assert (gb.grad.float() - ga.grad).abs().max() <= 2 ** -6 * ga.grad.abs().max()
    assert (ub.grad.float() - ua.grad).abs().max() <= 2 ** -6 * ua.grad.abs().max()


def test_scale_add_matches_addcmul():
    """LayerScale + residual (vaa_model_scale_add) == torch.addcmul bit for bit (one fp32 FMA, one rounding), and its autograd."""
    from roboticattack_amd import model_ops

    torch.manual_seed(6)
    x = torch.randn(3, 261, 1024, device=DEV).to(torch.bfloat16)
    a = torch.randn(3, 261, 1024, device=DEV).to(torch.bfloat16)
    ls = (0.1 * torch.randn(1024, device=DEV)).to(torch.bfloat16)
    xa, aa = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    xb, ab = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    ref = torch.addcmul(xa, aa, ls)
    got = model_ops.scale_add(xb, ab, ls)
    assert "ScaleAddFn" in type(got.grad_fn).__name__ and torch.equal(got, ref)
    g = torch.randn_like(ref)
    ref.backward(g)
    got.backward(g)
    assert torch.equal(xb.grad, xa.grad) and torch.equal(ab.grad, aa.grad)


def test_residual_rmsnorm_fwd_bwd_matches_eager():
    import torch.nn.functional as F

    from roboticattack_amd import model_ops

    torch.manual_seed(2)
    x = torch.randn(3, 70, 4096, device=DEV).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(4096, device=DEV)).to(torch.bfloat16)
    xa = x.float().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ref_h = F.rms_norm(xa, (4096,), w.float(), 1e-6)
    xo, h = model_ops.ResidualRMSNormFn.apply(xb, w, 1e-6)
    assert torch.equal(xo, xb) and (h.float() - ref_h).abs().max() <= 2 ** -6 * ref_h.abs().max()
    g_h, g_p = torch.randn_like(ref_h), torch.randn_like(ref_h)
    (ref_h * g_h + xa * g_p).sum().backward()  # residual stream used twice: through the norm and straight through
    (h.float() * g_h.to(torch.bfloat16).float() + xo.float() * g_p.to(torch.bfloat16).float()).sum().backward()
    assert (xb.grad.float() - xa.grad).abs().max() <= 2 ** -5 * xa.grad.abs().max()
    # norm output only (no pass-through gradient)
    xc = x.clone().requires_grad_(True)
    _, h2 = model_ops.ResidualRMSNormFn.apply(xc, w, 1e-6)
    h2.float().sum().backward()
    xd = x.float().requires_grad_(True)
    F.rms_norm(xd, (4096,), w.float(), 1e-6).sum().backward()
    assert (xc.grad.float() - xd.grad).abs().max() <= 2 ** -5 * xd.grad.abs().max() + 1e-3


def test_model_with_and_without_fused_ops(monkeypatch):
    """Same tiny-width OpenVLA-shaped model, bf16: fused RoPE/SwiGLU vs the eager chain — logits and pixel gradient agree to bf16 noise."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(32, 3, 2, 64, 5, True, True), siglip=VitCfg(48, 3, 2, 80, 0, False, False),
                     llm_dim=256, llm_layers=2, llm_heads=2, llm_mlp=512)  # head dim 128 like Llama-2
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=2)
    ids, labels, _ = synthetic.synth_text_batch(5, 2, 18, 22)
    labels = mask_labels(labels, [0]).to(DEV)
    pix0 = torch.randn(2, 6, 224, 224, device=DEV).to(torch.bfloat16)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("VAA_NO_FUSED_MODEL_OPS", "1")
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids.to(DEV), pix, labels)
        rows.float().square().mean().backward()
        outs.append((rows.detach().float(), pix.grad.detach().float()))
    (r1, g1), (r2, g2) = outs
    assert (r1 - r2).abs().max() <= 0.05 * r2.abs().max() + 1e-3
    assert torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0) > 0.99


def test_model_vision_towers_on_two_streams_bitwise(monkeypatch):
    """The SigLIP tower on a second HIP stream next to DINOv2 (default for per-rank batches <= 16): same kernels, same operands ->
    logits and pixel gradient are bitwise those of the single-stream run, repeatedly (no allocator reuse hazard across the streams)."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(128, 4, 2, 256, 5, False, True), siglip=VitCfg(192, 5, 2, 384, 0, False, False),
                     llm_dim=256, llm_layers=2, llm_heads=2, llm_mlp=512)
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=4)
    ids, labels, _ = synthetic.synth_text_batch(5, 4, 18, 22)
    labels = mask_labels(labels, [0]).to(DEV)
    pix0 = torch.randn(4, 6, 224, 224, device=DEV).to(torch.bfloat16)
    outs = []
    for mode in ("0", "1", "1", "0", "1"):
        monkeypatch.setenv("VAA_TOWER_STREAMS", mode)
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids.to(DEV), pix, labels)
        rows.float().square().mean().backward()
        junk = torch.randn(1 << 22, device=DEV)  # churn the caching allocator between runs
        del junk
        outs.append((rows.detach().clone(), pix.grad.detach().clone()))
    torch.cuda.synchronize()
    for r, g in outs[1:]:
        assert torch.equal(r, outs[0][0]) and torch.equal(g, outs[0][1])
    assert float(outs[0][1].float().abs().max()) > 0


def test_model_tn_dgrad_matches_autograd(monkeypatch):
    """Resident W^T + TN-layout dgrad (FrozenLinearsFn) vs autograd's `dy @ W`: same logits, same pixel gradient up to GEMM summation order."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(32, 3, 2, 64, 5, True, True), siglip=VitCfg(48, 3, 2, 80, 0, False, False),
                     llm_dim=256, llm_layers=2, llm_heads=2, llm_mlp=512)
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=4)
    ids, labels, _ = synthetic.synth_text_batch(6, 2, 18, 22)
    labels = mask_labels(labels, [0]).to(DEV)
    pix0 = torch.randn(2, 6, 224, 224, device=DEV).to(torch.bfloat16)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("VAA_NO_TN_DGRAD", "1")
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids.to(DEV), pix, labels)
        rows.float().square().mean().backward()
        outs.append((rows.detach().float(), pix.grad.detach().float()))
    (r1, g1), (r2, g2) = outs
    assert (r1 - r2).abs().max() <= 1e-2 * r2.abs().max() + 1e-4
    assert torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0) > 0.999
    # a weight update invalidates the cached transpose
    lyr = m.layers[0]
    wt_old = lyr._wt("q_proj")
    with torch.no_grad():
        lyr.q_proj.weight.mul_(2.0)
    assert torch.equal(lyr._wt("q_proj"), lyr.q_proj.weight.t()) and lyr._wt("q_proj") is not wt_old


@pytest.mark.parametrize("B,H,T,hd,causal,packed", [(2, 4, 300, 128, True, False), (2, 3, 261, 64, False, True), (2, 3, 256, 72, False, True),
                                                    (1, 2, 17, 128, True, False), (3, 2, 64, 64, True, True), (1, 9, 65, 72, False, True),
                                                    (2, 2, 130, 128, False, False), (1, 1, 1, 64, True, False), (1, 2, 333, 128, True, False),
                                                    (2, 2, 40, 16, True, False), (1, 2, 257, 24, False, True)])
def test_attention_fwd_bwd_vs_fp32(B, H, T, hd, causal, packed):
    """vaa_model_attention_{fwd,bwd} (MFMA, transposed formulation) vs fp32 softmax attention + autograd on the same bf16 inputs."""
    from roboticattack_amd import model_ops

    g = torch.Generator(device=DEV).manual_seed(T * 131 + hd)
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device=DEV, generator=g).to(torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = [(torch.randn(B, T, H, hd, device=DEV, generator=g) * 1.5).to(torch.bfloat16) for _ in range(3)]
    scale = hd ** -0.5
    go = torch.randn(B, T, H, hd, device=DEV, generator=g).to(torch.bfloat16)
    qf, kf, vf = [x.detach().float().requires_grad_(True) for x in (q, k, v)]
    s = torch.einsum("bthd,bshd->bhts", qf, kf) * scale
    if causal:
        s = s.masked_fill(torch.ones(T, T, device=DEV, dtype=torch.bool).triu(1), float("-inf"))
    ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vf)
    ref.backward(go.float())
    o, lse = model_ops.attention_fwd(q, k, v, causal, scale)
    assert (lse - torch.logsumexp(s, -1)).abs().max() < 1e-4
    assert (o.float() - ref).abs().max() <= 2 ** -7 * ref.abs().max() + 1e-3  # bf16 output rounding + bf16 P
    if packed:
        buf = model_ops.attention_bwd(q, k, v, o, lse, go, causal, scale, packed_grad=True)
        grads = (buf[:, :, 0], buf[:, :, 1], buf[:, :, 2])
    else:
        grads = model_ops.attention_bwd(q, k, v, o, lse, go, causal, scale)
    for d, r in zip(grads, (qf.grad, kf.grad, vf.grad)):
        assert (d.float() - r).abs().max() <= 2e-2 * r.abs().max() + 1e-5


def test_model_attention_matches_sdpa(monkeypatch):
    """The model with the matrix-core attention vs the same model on F.scaled_dot_product_attention (both bf16)."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(128, 3, 2, 256, 5, True, True), siglip=VitCfg(144, 3, 2, 288, 0, False, False),
                     llm_dim=256, llm_layers=2, llm_heads=2, llm_mlp=512)  # head dims 64 / 72 / 128 like the real towers
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=6)
    ids, labels, _ = synthetic.synth_text_batch(8, 2, 18, 22)
    labels = mask_labels(labels, [0]).to(DEV)
    pix0 = torch.randn(2, 6, 224, 224, device=DEV).to(torch.bfloat16)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("VAA_NO_FUSED_ATTENTION", "1")
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids.to(DEV), pix, labels)
        rows.float().square().mean().backward()
        outs.append((rows.detach().float(), pix.grad.detach().float()))
    (r1, g1), (r2, g2) = outs
    assert (r1 - r2).abs().max() <= 0.03 * r2.abs().max() + 1e-3
    assert torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0) > 0.995


@pytest.mark.parametrize("hd", [64, 128])
def test_rope_attention_fused_backward(hd):
    """RopeAttentionFn (rotary adjoint inside the attention backward epilogues) == RopeFn -> AttentionFn, up to one bf16 rounding."""
    from roboticattack_amd import model_ops

    B, T, H = 2, 75, 3
    g = torch.Generator(device=DEV).manual_seed(hd)
    ang = torch.outer(torch.arange(T, device=DEV, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd)))
    cos, sin = ang.cos().to(torch.bfloat16).float().contiguous(), ang.sin().to(torch.bfloat16).float().contiguous()
    base = [torch.randn(B, T, H, hd, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3)]
    go = torch.randn(B, T, H, hd, device=DEV, generator=g).to(torch.bfloat16)
    res = []
    for fused in (True, False):
        q, k, v = [x.clone().requires_grad_(True) for x in base]
        if fused:
            o = model_ops.RopeAttentionFn.apply(q, k, v, cos, sin, True, None)
        else:
            o = model_ops.AttentionFn.apply(model_ops.RopeFn.apply(q, cos, sin), model_ops.RopeFn.apply(k, cos, sin), v, True, None)
        o.backward(go)
        res.append((o.detach().float(), q.grad.float(), k.grad.float(), v.grad.float()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][3], res[1][3])
    for a, b in zip(res[0][1:3], res[1][1:3]):
        assert (a - b).abs().max() <= 2 ** -6 * b.abs().max()


@pytest.mark.parametrize("D", [1024, 1152, 48, 2048])  # one wave per row up to 1536 (2 / 3 vectors per lane), one workgroup per row beyond
def test_residual_layernorm_fwd_bwd_matches_eager(D):
    from roboticattack_amd import model_ops

    g = torch.Generator(device=DEV).manual_seed(D)
    x = (torch.randn(3, 37, D, device=DEV, generator=g) * 2 + 0.5).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(D, device=DEV, generator=g)).to(torch.bfloat16)
    b = (0.1 * torch.randn(D, device=DEV, generator=g)).to(torch.bfloat16)
    gp = torch.randn(3, 37, D, device=DEV, generator=g).to(torch.bfloat16)
    gh = torch.randn(3, 37, D, device=DEV, generator=g).to(torch.bfloat16)
    xa = x.clone().requires_grad_(True)
    xo, h = model_ops.ResidualLayerNormFn.apply(xa, w, b, 1e-6)
    (xo.float() * gp.float()).sum().backward(retain_graph=True)
    (h.float() * gh.float()).sum().backward()
    xb = x.float().clone().requires_grad_(True)
    hr = torch.nn.functional.layer_norm(xb, (D,), w.float(), b.float(), 1e-6)
    ((xb * gp.float()).sum() + (hr * gh.float()).sum()).backward()
    assert torch.equal(xo, x)
    assert (h.float() - hr).abs().max() <= 2 ** -7 * hr.abs().max()
    assert (xa.grad.float() - xb.grad).abs().max() <= 2 ** -6 * xb.grad.abs().max()


@pytest.mark.parametrize("hd,causal", [(128, True), (64, False)])
def test_attention_packed_sequences_match_per_sample(hd, causal):
    """cu_seqlens form (sequences packed back to back, no padding rows) == running every sample on its own, forward and backward,
    including the fused rotary adjoint with per-token tables."""
    from roboticattack_amd import model_ops

    lens = [37, 64, 101, 1, 130]
    H = 2
    g = torch.Generator(device=DEV).manual_seed(hd)
    tot = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    q, k, v, go = [torch.randn(1, tot, H, hd, device=DEV, generator=g).to(torch.bfloat16) for _ in range(4)]
    pos = torch.cat([torch.arange(n, device=DEV) for n in lens])
    ang = torch.outer(torch.arange(max(lens), device=DEV, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd)))
    cos, sin = ang.cos().to(torch.bfloat16).float().contiguous(), ang.sin().to(torch.bfloat16).float().contiguous()
    cos_p, sin_p = cos.index_select(0, pos).contiguous(), sin.index_select(0, pos).contiguous()
    qa, ka, va = [x.clone().requires_grad_(True) for x in (q, k, v)]
    if causal:
        o = model_ops.RopeAttentionFn.apply(qa, ka, va, cos_p, sin_p, True, None, cu, max(lens))
    else:
        o, lse = model_ops.attention_fwd(qa, ka, va, False, None, cu, max(lens))
    outs, grads = [], []
    for i, n in enumerate(lens):
        s0 = int(cu[i])
        qi, ki, vi = [x[:, s0:s0 + n].clone().requires_grad_(True) for x in (q, k, v)]
        if causal:
            oi = model_ops.RopeAttentionFn.apply(qi, ki, vi, cos[:n].contiguous(), sin[:n].contiguous(), True, None)
            oi.backward(go[:, s0:s0 + n])
            grads.append((qi.grad, ki.grad, vi.grad))
        else:
            oi, _ = model_ops.attention_fwd(qi, ki, vi, False, None)
        outs.append(oi.detach())
    assert torch.equal(o.detach(), torch.cat(outs, dim=1))
    if causal:
        o.backward(go)
        for j, ga in enumerate((qa.grad, ka.grad, va.grad)):
            assert torch.equal(ga, torch.cat([gr[j] for gr in grads], dim=1))


def test_model_sequence_packing_matches_padded(monkeypatch):
    """SeqPack (padding rows never computed) gives the same labelled-row logits and the same pixel gradient as the padded batch."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(32, 3, 2, 64, 5, True, True), siglip=VitCfg(48, 3, 2, 80, 0, False, False),
                     llm_dim=256, llm_layers=3, llm_heads=2, llm_mlp=512)
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=12)
    batch = synthetic.synth_batch(11, 5, "noise", as_pil=False)
    ids, attn = batch["input_ids"].to(DEV), batch["attention_mask"].to(DEV)
    labels = mask_labels(batch["labels"].clone(), [0, 3]).to(DEV)
    assert int(attn.sum()) < attn.numel()  # there IS padding to drop
    assert m.make_pack(attn) is None  # opt-in
    monkeypatch.setenv("VAA_SEQ_PACK", "1")
    pack = m.make_pack(attn)
    assert pack is not None and pack.total == int(attn.sum()) + 5 * 256
    pix0 = torch.randn(5, 6, 224, 224, device=DEV).to(torch.bfloat16)
    res = []
    for p in (pack, None):
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids, pix, labels, pack=p)
        rows.float().square().mean().backward()
        res.append((rows.detach().float(), pix.grad.detach().float()))
    (r1, g1), (r2, g2) = res
    assert (r1 - r2).abs().max() <= 2e-2 * r2.abs().max() + 1e-3   # same math; GEMM row count differs -> different kernel selections
    assert torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0) > 0.999


@pytest.mark.gpu
@pytest.mark.parametrize("hd", [64, 128])
def test_rope_packed_attention_equals_separate_operands(hd):
    """RopePackedAttentionFn on one [B,T,3,H,hd] buffer (a fused q/k/v projection's output) == RopeAttentionFn on three tensors, bit for bit:
    output and the packed gradient's dq | dk | dv slices (same kernels, strided operands)."""
    from roboticattack_amd import model_ops

    B, T, H = 2, 75, 3
    g = torch.Generator(device=DEV).manual_seed(hd + 1)
    ang = torch.outer(torch.arange(T, device=DEV, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd)))
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    qkv0 = torch.randn(B, T, 3, H, hd, device=DEV, generator=g).to(torch.bfloat16)
    go = torch.randn(B, T, H, hd, device=DEV, generator=g).to(torch.bfloat16)
    qkv = qkv0.clone().requires_grad_(True)
    o1 = model_ops.RopePackedAttentionFn.apply(qkv, cos, sin, True, None)
    o1.backward(go)
    q, k, v = [qkv0[:, :, z].contiguous().requires_grad_(True) for z in range(3)]
    o2 = model_ops.RopeAttentionFn.apply(q, k, v, cos, sin, True, None)
    o2.backward(go)
    assert torch.equal(o1, o2)
    for z, t in enumerate((q, k, v)):
        assert torch.equal(qkv.grad[:, :, z], t.grad)


@pytest.mark.gpu
def test_model_fused_qkv_matches_separate_projections(monkeypatch):
    """q/k/v as ONE GEMM over the concatenated weights (the default below 8,192 rows) against three GEMMs: same logits rows and pixel gradient up
    to GEMM summation order; the concatenated copy follows a weight update."""
    from roboticattack_amd import synthetic
    from roboticattack_amd.labels import mask_labels
    from roboticattack_amd.openvla_model import OpenVLACfg, VitCfg, build_openvla

    cfg = OpenVLACfg(dino=VitCfg(32, 3, 2, 64, 5, True, True), siglip=VitCfg(48, 3, 2, 80, 0, False, False),
                     llm_dim=256, llm_layers=2, llm_heads=2, llm_mlp=512)  # head dim 128
    m = build_openvla(cfg, device=DEV, dtype=torch.bfloat16, seed=4)
    ids, labels, _ = synthetic.synth_text_batch(6, 2, 18, 22)
    labels = mask_labels(labels, [0]).to(DEV)
    pix0 = torch.randn(2, 6, 224, 224, device=DEV).to(torch.bfloat16)
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("VAA_FUSED_QKV", mode)
        pix = pix0.clone().requires_grad_(True)
        rows = m.forward_rows(ids.to(DEV), pix, labels)
        rows.float().square().mean().backward()
        outs.append((rows.detach().float(), pix.grad.detach().float()))
    (r1, g1), (r2, g2) = outs
    assert (r1 - r2).abs().max() <= 1e-2 * r2.abs().max() + 1e-4
    assert torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0) > 0.999
    monkeypatch.setenv("VAA_FUSED_QKV", "1")
    lyr = m.layers[0]
    w_old, _ = lyr._wcat(("q_proj", "k_proj", "v_proj"))
    with torch.no_grad():
        lyr.k_proj.weight.mul_(2.0)
    w_new, wt_new = lyr._wcat(("q_proj", "k_proj", "v_proj"))
    assert w_new is not w_old and torch.equal(w_new[256:512], lyr.k_proj.weight) and torch.equal(wt_new, w_new.t())


def test_attention_two_row_groups_per_wave_matches_fp32():
    """The attention kernels with TWO 16-row groups per wave (the default only for the 72-wide SigLIP heads; VAA_ATTN_G forces it for every
    head width) against fp32 softmax attention at the three shapes of the step — a separate process, because the grouping is read once."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cfg in ("222", "111"):
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_bench.py"), "--check", "--bs", "2", "--iters", "1"],
                           env=dict(os.environ, VAA_ATTN_G=cfg), cwd=root, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        assert p.stdout.count('"ok": true') == 3, p.stdout
