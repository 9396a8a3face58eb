"""GPU parity tests of K3s (vaa_head_slice_fwd_bwd): the slice-only LM head + statistics + gradient + head backward in one launch.

Bars: action logits, slice statistics, UAD, loss scalars and the gradient slice BIT FOR BIT those of K3h (vaa_head_loss_rows_stats + finish) on the same
hidden rows and weight — the kernel runs the same MFMA sequence per logit; the C oracle on those logits; dH against the exact (fp64) products of the
bf16 gradient slice with the bf16 weight slice within half a bf16 step + an fp32 accumulation allowance.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from roboticattack_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
V = 32064


@pytest.fixture(scope="module")
def ops():
    from roboticattack_amd import ops as _ops

    _ops.device_check()
    return _ops


def _rows(labels):
    """(b, model position) of every labelled position, (b,k) row-major: row (b, 256 + k) predicts labels[b, k+1]."""
    B, L = labels.shape
    return [(b, 256 + k) for b in range(B) for k in range(L - 1) if labels[b, k + 1] != -100]


def _case(ops, B, maskidx, D, seed, upa=False):
    from roboticattack_amd.labels import mask_labels

    _, labels, _ = synthetic.synth_text_batch(seed, B)
    if not upa:  # UPA's reverse-direction mode keeps all eight labelled positions (UPA.py:127-129)
        labels = mask_labels(labels, maskidx)
    g = torch.Generator(device=DEV).manual_seed(seed * 3 + D)
    W = (torch.randn(V, D, device=DEV, generator=g) * (1.3 / np.sqrt(D))).to(torch.bfloat16)
    W[31744:32000] *= 2.0  # spread the action logits
    R = len(_rows(labels.numpy()))
    h = torch.randn(R, D, device=DEV, generator=g).to(torch.bfloat16)
    return labels, W, h, ops.LossRowMap(labels.to(DEV)), R


def _bits(t):
    return t.contiguous().view(torch.int16)


@pytest.mark.parametrize("mode_name", ["UADA_DDP", "UPA"])
@pytest.mark.parametrize("B,maskidx,D", [(4, [0], 4096), (8, [0], 4096), (32, [0], 4096), (64, [0], 4096), (13, [0, 1, 2], 4096), (5, [0, 3], 192), (7, [0, 1, 2], 320),
                                         (3, [6], 64), (9, [0, 1, 2, 3, 4, 5, 6], 1024)])
def test_head_slice_equals_k3h_bitwise_and_oracle(ops, mode_name, B, maskidx, D):
    upa = mode_name == "UPA"
    if upa and B * 8 > 128:
        pytest.skip("UPA keeps 8 rows per sample: more than 128 rows")
    mode = ops.LOSS_UPA if upa else ops.LOSS_UADA_DDP
    labels, W, h, rm, R = _case(ops, B, maskidx, D, 700 + B, upa)
    L = labels.shape[1]
    assert ops.head_slice_applies(R, D, V) and ops.head_loss_rows_applies(R, D, V)
    kw = dict(w=4.0, alpha=0.7, beta=0.3, scale=1.0)
    # K3h: full-vocabulary stream + finish (the path of every step until round 5)
    sc, pred, pf, gs, lg = ops.head_loss_rows_fwd_bwd(h, W, rm, mode, want_grad=True, want_logits=True, **kw)
    sc, pred, pf, gs = sc.clone(), pred.clone(), pf.clone(), gs.clone()
    # K3s
    o = ops.head_slice_fwd_bwd(h, W, rm, mode, want_dh=True, want_scalars=True, want_grad_slice=True, **kw)
    torch.cuda.synchronize()
    ops.async_error_check()
    words = o["zs"][: R * 1024].view(torch.int64).view(R, 128)  # {launch tag : 32 | logit 2q+1 : 16 | logit 2q : 16}
    assert int((words >> 32).unique().numel()) == 1
    zs = (words & 0xFFFFFFFF).to(torch.int32).view(torch.bfloat16).view(R, 256)
    assert torch.equal(_bits(zs), _bits(lg[:, 31744:32000])), "action logits differ from K3h's"
    assert torch.equal(_bits(o["grad_slice"]), _bits(gs)) and float(gs.float().abs().max()) > 0
    s2 = o["scalars"]
    assert torch.equal(s2[[0, 2, 3, 4, 5, 6, 7]], sc[[0, 2, 3, 4, 5, 6, 7]]), (s2, sc)  # total, w^2 MSE, UPA angle / dist, row counts, UAD
    assert float(s2[1]) == 0.0 and float(sc[1]) > 0.0                                     # CE: not evaluated on a slice-only step
    assert torch.equal(o["pred"], pred)
    assert int((o["pred_full"] != -1).sum()) == 0 and int((pf >= 0).sum()) == R
    # the oracle on K3h's logits (the same bits as K3s's in the action columns)
    rows = _rows(labels.numpy())
    rb, rp = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
    full = torch.zeros((B, 256 + L, V), dtype=torch.float32)
    full[torch.from_numpy(rb), torch.from_numpy(rp)] = lg.float().cpu()
    so, go = c_oracle.loss(full.numpy(), labels.numpy(), c_oracle.MODE_UPA if upa else c_oracle.MODE_UADA_DDP, **kw)
    got = s2.cpu().numpy()
    assert np.allclose(got[[0, 2, 3, 4]], so[[0, 2, 3, 4]], rtol=3e-5, atol=3e-5), (got, so)
    gor = go[rb, rp][:, 31744:32000]
    assert np.abs(o["grad_slice"].float().cpu().numpy() - gor).max() <= 1e-2 * max(np.abs(gor).max(), 1e-30)
    # dH = g (bf16) x W[31744:32000] (bf16): exact products within half a bf16 step + an fp32 accumulation of 256 terms
    g64, w64 = gs.double(), W[31744:32000].double()
    ref, mag = g64 @ w64, g64.abs() @ w64.abs()
    dh = o["dh"].double()
    assert bool(((dh - ref).abs() <= 0.5 * torch.maximum(ref.abs(), dh.abs()) * 2.0 ** -7 + 2e-6 * mag + 1e-30).all())
    assert float(dh.abs().max()) > 0
    # ... and the library GEMM the step used until now agrees to bf16 rounding
    lib = (gs @ W[31744:32000]).double()
    assert float((lib - dh).abs().max()) <= 2.0 ** -6 * float(ref.abs().max())
    # bitwise repeatable, and the two-launch form leaves the same bits
    o2 = ops.head_slice_fwd_bwd(h, W, rm, mode, want_dh=True, want_scalars=True, want_grad_slice=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(_bits(o2["dh"]), _bits(o["dh"])) and torch.equal(o2["scalars"], s2)


@pytest.mark.parametrize("mode_name,B,maskidx", [("UADA_DDP", 64, [0]), ("UADA_DDP", 8, [0, 1, 2]), ("UPA", 4, [0, 1, 2]), ("UPA", 16, [0])])
def test_head_slice_two_launches_equal_one(ops, monkeypatch, mode_name, B, maskidx):
    upa = mode_name == "UPA"
    mode = ops.LOSS_UPA if upa else ops.LOSS_UADA_DDP
    labels, W, h, rm, R = _case(ops, B, maskidx, 4096, 1300 + B, upa)
    a = ops.head_slice_fwd_bwd(h, W, rm, mode, want_grad_slice=True)
    a = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in a.items()}
    monkeypatch.setenv("VAA_K3S_ONE_LAUNCH", "0")
    b = ops.head_slice_fwd_bwd(h, W, rm, mode, want_grad_slice=True)
    torch.cuda.synchronize()
    for k in ("dh", "grad_slice"):
        assert torch.equal(_bits(a[k]), _bits(b[k])), k
    assert torch.equal(a["scalars"], b["scalars"]) and torch.equal(a["pred"], b["pred"])


@pytest.mark.parametrize("mode_name,B,maskidx,D", [("UADA_DDP", 64, [0], 4096), ("UADA_DDP", 8, [0], 4096), ("UADA_DDP", 13, [0, 1, 2], 4096), ("UPA", 16, [0, 1, 2], 4096),
                                                   ("UPA", 4, [0, 1, 2], 4096), ("UADA_DDP", 5, [0, 3], 192), ("UPA", 7, [0, 1, 2], 320)])
def test_head_slice_forms_agree_bit_for_bit(ops, monkeypatch, mode_name, B, maskidx, D):
    """The kernel's forms — 8 action columns per workgroup (32 workgroups per row block: the default), 16 columns (VAA_K3S_COLS=16), and 16 columns with
    row-block groups of 2 / 4 (VAA_K3S_GROUP) — are the same arithmetic on a different decomposition: every output bit for bit, one launch or two."""
    upa = mode_name == "UPA"
    mode = ops.LOSS_UPA if upa else ops.LOSS_UADA_DDP
    labels, W, h, rm, R = _case(ops, B, maskidx, D, 4400 + B, upa)
    for k in ("VAA_K3S_COLS", "VAA_K3S_GROUP", "VAA_K3S_ONE_LAUNCH"):  # (a run of the whole file under one of the switches)
        monkeypatch.delenv(k, raising=False)

    def run():
        o = ops.head_slice_fwd_bwd(h, W, rm, mode, want_grad_slice=True)
        torch.cuda.synchronize()
        return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in o.items()}

    ops.prof_start(64)
    a = run()
    names = [n for n, _ in ops.prof_collect()]
    assert any("head_slice_kernel<1,8>" in n for n in names), names  # the default IS the 8-column form
    for env in ({"VAA_K3S_COLS": "16"}, {"VAA_K3S_COLS": "16", "VAA_K3S_ONE_LAUNCH": "0"}, {"VAA_K3S_ONE_LAUNCH": "0"}, {"VAA_K3S_GROUP": "2"}, {"VAA_K3S_GROUP": "4"}):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            ops.prof_start(64)
            b = run()
            names = [n for n, _ in ops.prof_collect()]
        if "VAA_K3S_COLS" in env:
            assert any("head_slice_kernel<1>" in n for n in names) and not any("<1,8>" in n for n in names), names
        for k in ("dh", "grad_slice"):
            assert torch.equal(_bits(a[k]), _bits(b[k])), (env, k)
        assert torch.equal(a["scalars"], b["scalars"]) and torch.equal(a["pred"], b["pred"]) and torch.equal(a["pred_full"], b["pred_full"]), env


@pytest.mark.parametrize("B,maskidx", [(64, [0]), (8, [0]), (5, [0, 1, 2])])
def test_head_slice_with_step_epilogue_and_ce_steps(ops, B, maskidx):
    """The data-parallel step's cadence: K3s every step (its SliceStats + neutral parts folded by vaa_step_epilogue: CE = 0, pred_full = -1), and on the
    steps whose CE is read K3h's statistics pass AFTER it on the same workspace: the epilogue then reports K3h's CE / full argmax and the same slice scalars."""
    labels, W, h, rm, R = _case(ops, B, maskidx, 4096, 2100 + B)
    n = 7500
    parts, msg = torch.zeros((4, n), device=DEV), torch.zeros(n + 4, device=DEV)
    # reference: K3h alone
    gs = torch.empty((R, 256), dtype=torch.bfloat16, device=DEV)
    ws = ops.head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs)
    sc_h = torch.zeros(8, device=DEV)
    pred_h, pf_h = ops.step_epilogue(parts, msg, sc_h, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws)
    sc_h, pred_h, pf_h, tail_h = sc_h.clone(), pred_h.clone(), pf_h.clone(), msg[n:].clone()
    # slice-only step
    o = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=False, want_grad_slice=True)
    sc_s = torch.zeros(8, device=DEV)
    pred_s, pf_s = ops.step_epilogue(parts, msg, sc_s, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=o["ws"])
    torch.cuda.synchronize()
    assert torch.equal(_bits(o["grad_slice"]), _bits(gs))
    assert torch.equal(sc_s[[0, 2, 5, 6, 7]], sc_h[[0, 2, 5, 6, 7]]) and float(sc_s[1]) == 0.0
    assert torch.equal(pred_s, pred_h) and int((pf_s != -1).sum()) == 0
    assert torch.equal(msg[n + 1 :], tail_h[1:]) and float(msg[n]) == 0.0
    # CE step: K3s, then K3h's statistics on the same workspace
    o = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP, 5.0, want_scalars=False)
    ws2 = ops.head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0)
    assert ws2.data_ptr() == o["ws"].data_ptr()
    sc_c = torch.zeros(8, device=DEV)
    pred_c, pf_c = ops.step_epilogue(parts, msg, sc_c, rowmap=rm, R=R, V=V, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws2)
    torch.cuda.synchronize()
    assert torch.equal(sc_c, sc_h) and torch.equal(pred_c, pred_h) and torch.equal(pf_c, pf_h) and torch.equal(msg[n:], tail_h)


@pytest.mark.parametrize("mode_name,B", [("UADA_DDP", 16), ("UPA", 8)])
def test_head_slice_forward_only(ops, mode_name, B):
    """Validation passes (no gradient): scalars and the slice argmax map of the full call, nothing else written."""
    upa = mode_name == "UPA"
    mode = ops.LOSS_UPA if upa else ops.LOSS_UADA_DDP
    labels, W, h, rm, R = _case(ops, B, [0, 1, 2], 4096, 3100 + B, upa)
    a = ops.head_slice_fwd_bwd(h, W, rm, mode)
    sa, pa = a["scalars"].clone(), a["pred"].clone()
    b = ops.head_slice_fwd_bwd(h, W, rm, mode, want_dh=False)
    torch.cuda.synchronize()
    assert b["dh"] is None and torch.equal(b["scalars"], sa) and torch.equal(b["pred"], pa)


def test_head_slice_handover_timeout_fails_loudly(ops, monkeypatch):
    """VAA_K3_HANDOVER_POLLS=0: every waiting workgroup gives up at once — dH comes back NaN-poisoned (never stale) and the library's failure word is
    raised; the poll clears it and the next call is clean."""
    from roboticattack_amd import _lib

    labels, W, h, rm, R = _case(ops, 64, [0], 4096, 4100)
    good = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP)["dh"].clone()
    monkeypatch.setenv("VAA_K3_HANDOVER_POLLS", "0")
    try:
        bad = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP)
        torch.cuda.synchronize()
        raised = False
        try:
            ops.async_error_check()
        except _lib.VaaError:
            raised = True
        # a workgroup that arrives last finds its own word at once: at least the others must have given up
        if raised:
            assert bool(torch.isnan(bad["dh"].float()).any())
    except _lib.VaaError:  # the sticky word was seen by the call's own check_launch
        torch.cuda.synchronize()
        try:
            ops.async_error_check()
        except _lib.VaaError:
            pass
    finally:
        monkeypatch.delenv("VAA_K3_HANDOVER_POLLS")
        try:
            ops.async_error_check()
        except _lib.VaaError:
            pass
    again = ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA_DDP)["dh"]
    torch.cuda.synchronize()
    ops.async_error_check()
    assert torch.equal(_bits(again), _bits(good))


def test_head_slice_argument_checks(ops):
    from roboticattack_amd import _lib

    labels, W, h, rm, R = _case(ops, 4, [0], 256, 5100)
    L = _lib.lib()
    assert L.vaa_head_slice_applies(128, 4096, V) == 1 and L.vaa_head_slice_applies(129, 4096, V) == 0
    assert L.vaa_head_slice_applies(16, 4160, V) == 0 and L.vaa_head_slice_applies(16, 96, V) == 0 and L.vaa_head_slice_applies(16, 64, 40000) == 0
    assert L.vaa_head_slice_ws_bytes(17) == 32 * 1024 and L.vaa_head_slice_ws_bytes(40) == 48 * 1024 and L.vaa_head_slice_ws_bytes(128) == 128 * 1024 and L.vaa_head_slice_ws_bytes(0) == 0
    with pytest.raises(_lib.VaaError, match="cross-entropy"):
        ops.head_slice_fwd_bwd(h, W, rm, ops.LOSS_UADA)
    p = _lib.f32x([5.0, 0.8, 0.2, 1.0])
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    rc = L.vaa_head_slice_fwd_bwd(h.data_ptr(), W.data_ptr(), None, 256, rm.buf.data_ptr(), R, rm.B, rm.L, V, ops.LOSS_UADA_DDP, p, ws.data_ptr(), None,
                                  ws.data_ptr(), ws.numel(), None, None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"null pointer" in L.vaa_last_error()  # a gradient without the transposed slice
    rc = L.vaa_head_slice_fwd_bwd(h.data_ptr(), W.data_ptr(), None, 256, rm.buf.data_ptr(), R, rm.B, rm.L, V, ops.LOSS_UADA_DDP, p, None, None,
                                  ws.data_ptr(), 16, None, None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -4  # VAA_E_WORKSPACE
    rc = L.vaa_head_slice_pack(None, 256, V, None, None)
    assert rc == -1
    # the transposed slice is the slice
    wt = ops.head_slice_packed(W)
    torch.cuda.synchronize()
    assert torch.equal(_bits(wt), _bits(W[31744:32000].t().contiguous()))
    assert ops.head_slice_packed(W) is wt  # cached per weight tensor
