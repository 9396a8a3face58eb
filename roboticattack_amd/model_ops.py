"""Optional fused elementwise operators for the PyTorch-ROCm model (include/vaa_model_ops.h): RoPE, SwiGLU, the norms, LayerScale + residual, attention.

Outside the hot-path contract (SURVEY.md §8a-5 keeps the model as stock PyTorch): they only shave the eager elementwise
chains that rocprofv3 shows around the GEMMs. `enabled(x)` is False for non-bf16 / non-ROCm tensors or when
VAA_NO_FUSED_MODEL_OPS=1, in which case openvla_model.py runs the plain PyTorch formulation.
"""
from __future__ import annotations

import os

import torch

from . import _lib


def enabled(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.bfloat16 and not os.environ.get("VAA_NO_FUSED_MODEL_OPS")


def attention_enabled(x_bthd: torch.Tensor) -> bool:
    """Matrix-core attention (vaa_attention.hip) applies: bf16 on ROCm, head dim % 8 == 0 and <= 128, 16-byte aligned strides."""
    hd = x_bthd.shape[-1]
    return (enabled(x_bthd) and not os.environ.get("VAA_NO_FUSED_ATTENTION") and hd % 8 == 0 and hd <= 128
            and all(s % 8 == 0 for s in x_bthd.stride()[:-1]) and x_bthd.stride(-1) == 1)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rope_launch(x_bthd: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, sign: float) -> torch.Tensor:
    B, T, H, hd = x_bthd.shape
    if x_bthd.stride(3) != 1:
        x_bthd = x_bthd.contiguous()
    out = torch.empty((B, T, H, hd), dtype=torch.bfloat16, device=x_bthd.device)
    rc = _lib.lib().vaa_model_rope(x_bthd.data_ptr(), x_bthd.stride(0), x_bthd.stride(1), x_bthd.stride(2), cos.data_ptr(), sin.data_ptr(),
                                   B, T, H, hd, float(sign), out.data_ptr(), _stream())
    _lib.check(rc, "vaa_model_rope")
    return out


class RopeFn(torch.autograd.Function):
    """x [B,T,H,hd] bf16 (any b/t/h strides), cos/sin float32 [T,hd/2] -> rotated, contiguous [B,T,H,hd]."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        ctx.save_for_backward(cos, sin)
        return _rope_launch(x, cos, sin, 1.0)

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        return _rope_launch(g, cos, sin, -1.0), None, None


class SwiGLUFn(torch.autograd.Function):
    """silu(gate) * up, bf16, contiguous; backward in one pass."""

    @staticmethod
    def forward(ctx, gate, up):
        gate, up = gate.contiguous(), up.contiguous()
        y = torch.empty_like(gate)
        _lib.check(_lib.lib().vaa_model_swiglu_fwd(gate.data_ptr(), up.data_ptr(), y.data_ptr(), gate.numel(), _stream()), "vaa_model_swiglu_fwd")
        ctx.save_for_backward(gate, up)
        return y

    @staticmethod
    def backward(ctx, dy):
        gate, up = ctx.saved_tensors
        dy = dy.contiguous()
        dg, du = torch.empty_like(gate), torch.empty_like(up)
        _lib.check(_lib.lib().vaa_model_swiglu_bwd(dy.data_ptr(), gate.data_ptr(), up.data_ptr(), dg.data_ptr(), du.data_ptr(), gate.numel(), _stream()),
                   "vaa_model_swiglu_bwd")
        return dg, du


def _scale_add(x, a, ls):
    D = a.shape[-1]
    out = torch.empty_like(a)
    _lib.check(_lib.lib().vaa_model_scale_add(x.data_ptr() if x is not None else None, a.data_ptr(), ls.data_ptr(), out.data_ptr(), a.numel() // D, D,
                                              _stream()), "vaa_model_scale_add")
    return out


class ScaleAddFn(torch.autograd.Function):
    """x + a * ls (LayerScale + residual, ls [D] frozen): one streaming kernel; backward: dx = g (passed through), da = g * ls."""

    @staticmethod
    def forward(ctx, x, a, ls):
        ctx.save_for_backward(ls)
        return _scale_add(x.contiguous(), a.contiguous(), ls.contiguous())

    @staticmethod
    def backward(ctx, g):
        (ls,) = ctx.saved_tensors
        g = g.contiguous()
        da = _scale_add(None, g, ls.contiguous()) if ctx.needs_input_grad[1] else None
        return (g if ctx.needs_input_grad[0] else None), da, None  # (ls is frozen: scale_add dispatches here only then)


def scale_add(x: torch.Tensor, a: torch.Tensor, ls: torch.Tensor) -> torch.Tensor:
    """torch.addcmul(x, a, ls) for a LayerScale vector ls [D]; the fused kernel for bf16 ROCm tensors of equal shape, D % 8 == 0."""
    if (enabled(x) and a.dtype == x.dtype and ls.dtype == x.dtype and x.shape == a.shape and ls.dim() == 1 and ls.shape[0] == x.shape[-1]
            and x.shape[-1] % 8 == 0 and a.device == x.device and ls.device == x.device and not ls.requires_grad  # (an unfrozen LayerScale needs addcmul's gradient)
            and os.environ.get("VAA_MODEL_SCALE_ADD", "1") != "0"):
        return ScaleAddFn.apply(x, a, ls)
    return torch.addcmul(x, a, ls)


class ResidualRMSNormFn(torch.autograd.Function):
    """(x) -> (x, rmsnorm(x) * w): the residual stream passes through untouched, so autograd hands BOTH incoming gradients to
    one backward kernel (no separate grad-accumulation add, no multi-kernel norm backward). The weight is frozen (no grad)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        h = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().vaa_model_rmsnorm_fwd(x.data_ptr(), weight.data_ptr(), h.data_ptr(), rstd.data_ptr(), rows, D, float(eps), _stream()),
                   "vaa_model_rmsnorm_fwd")
        ctx.save_for_backward(x, weight, rstd)
        return x, h

    @staticmethod
    def backward(ctx, g_pass, g_h):
        x, weight, rstd = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        if g_h is None:
            return g_pass, None, None
        g_h = g_h.contiguous()
        gp = g_pass.contiguous() if g_pass is not None else None
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().vaa_model_rmsnorm_bwd(g_h.data_ptr(), gp.data_ptr() if gp is not None else None, x.data_ptr(), weight.data_ptr(),
                                                    rstd.data_ptr(), gx.data_ptr(), rows, D, _stream()), "vaa_model_rmsnorm_bwd")
        return gx, None, None


class ResidualLayerNormFn(torch.autograd.Function):
    """(x) -> (x, layer_norm(x) * w + b) for the ViT blocks: same pass-through trick as ResidualRMSNormFn (one backward kernel
    receives the residual-stream gradient and the branch gradient). Weight and bias are frozen (no grads)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        h = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().vaa_model_layernorm_fwd(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), h.data_ptr(), stats.data_ptr(), rows, D,
                                                      float(eps), _stream()), "vaa_model_layernorm_fwd")
        ctx.save_for_backward(x, weight, stats)
        return x, h

    @staticmethod
    def backward(ctx, g_pass, g_h):
        x, weight, stats = ctx.saved_tensors
        D = x.shape[-1]
        rows = x.numel() // D
        if g_h is None:
            return g_pass, None, None, None
        g_h = g_h.contiguous()
        gp = g_pass.contiguous() if g_pass is not None else None
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().vaa_model_layernorm_bwd(g_h.data_ptr(), gp.data_ptr() if gp is not None else None, x.data_ptr(), weight.data_ptr(),
                                                      stats.data_ptr(), gx.data_ptr(), rows, D, _stream()), "vaa_model_layernorm_bwd")
        return gx, None, None, None


def tn_dgrad_enabled() -> bool:
    return not os.environ.get("VAA_NO_TN_DGRAD")


class FrozenLinearsFn(torch.autograd.Function):
    """(x, res, W_0, Wt_0, W_1, Wt_1, ...) -> tuple of x @ W_i^T (+ res for a single weight, fused as the GEMM epilogue).

    The weights are frozen, so the backward is data-gradient only: dx = sum_i dy_i @ W_i. Autograd's `dy @ W` is an NN-layout
    GEMM, which hipBLASLt runs 20-25 % slower on gfx950 than the TN layout of the forward (tools/gemm_layout.py:
    19200x11008x4096 1440 vs 1134 us). With 288 GB of HBM the transposed copy W_i^T (`Wt_i`, [in,out] contiguous) of every
    Llama projection is simply kept resident (+12.9 GB), which turns each dgrad into a TN GEMM of an already-tuned forward
    shape; the sum over i runs in the GEMM epilogue (addmm beta=1) instead of autograd's separate accumulation kernels."""

    @staticmethod
    def forward(ctx, x, res, *ws):
        x2 = x.reshape(-1, x.shape[-1])
        W, Wt = ws[0::2], ws[1::2]
        ctx.wt = Wt
        ctx.xshape = x.shape
        ctx.has_res = res is not None
        if res is not None:
            return (torch.addmm(res.reshape(-1, W[0].shape[0]), x2, W[0].t()).view(*x.shape[:-1], W[0].shape[0]),)
        return tuple(torch.nn.functional.linear(x2, w).view(*x.shape[:-1], w.shape[0]) for w in W)

    @staticmethod
    def backward(ctx, *dys):
        dx = None
        for dy, wt in zip(dys, ctx.wt):
            if dy is None:
                continue
            d2 = dy.reshape(-1, dy.shape[-1])
            if dx is None:
                dx = torch.nn.functional.linear(d2, wt)  # [M,out] x [in,out]^T
            else:
                dx.addmm_(d2, wt.t())
        dx = dx.view(ctx.xshape) if dx is not None else None
        dres = dys[0] if ctx.has_res else None
        return (dx, dres) + (None,) * (2 * len(ctx.wt))


def _str3(x: torch.Tensor):
    """Element strides {batch, token, head} of a [B,T,H,hd] view as a C int64[3]."""
    import ctypes

    if x.stride(3) != 1:
        raise ValueError("attention operands need a contiguous head dimension")
    return (ctypes.c_int64 * 3)(x.stride(0), x.stride(1), x.stride(2))


def attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool, scale: float | None = None, cu_seqlens=None, max_len: int = 0):
    """softmax(scale * q k^T [causal]) v for bf16 [B,T,H,hd] views (any batch/token/head strides); returns (o [B,T,H,hd]
    contiguous, lse float32 [B,H,T]). Packed form: q/k/v are [1,sumT,H,hd], cu_seqlens int32 [B+1] on the device, max_len = longest
    sequence; o is [1,sumT,H,hd] and lse [B,H,max_len]."""
    Bq, T, H, hd = q.shape
    scale = float(hd) ** -0.5 if scale is None else float(scale)
    o = torch.empty((Bq, T, H, hd), dtype=torch.bfloat16, device=q.device)
    B, Tm = (int(cu_seqlens.numel()) - 1, int(max_len)) if cu_seqlens is not None else (Bq, T)
    lse = torch.empty((B, H, Tm), dtype=torch.float32, device=q.device)
    rc = _lib.lib().vaa_model_attention_fwd(q.data_ptr(), _str3(q), k.data_ptr(), _str3(k), v.data_ptr(), _str3(v), o.data_ptr(), _str3(o),
                                            lse.data_ptr(), cu_seqlens.data_ptr() if cu_seqlens is not None else None, B, H, Tm, hd,
                                            int(bool(causal)), scale, _stream())
    _lib.check(rc, "vaa_model_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, lse, dout, causal: bool, scale: float, packed_grad: bool = False, rope=None, cu_seqlens=None, max_len: int = 0):
    """Gradients of attention_fwd. packed_grad=True returns one [B,T,3,H,hd] buffer (dq|dk|dv slices) — the layout of a fused
    qkv projection's output gradient — instead of three [B,T,H,hd] tensors. rope=(cos, sin) float32 [T,hd/2]: q and k are the
    ROTATED tensors and dq/dk come back w.r.t. the un-rotated ones (adjoint rotation fused into the kernels' epilogues)."""
    Bq, T, H, hd = q.shape
    B, Tm = (int(cu_seqlens.numel()) - 1, int(max_len)) if cu_seqlens is not None else (Bq, T)
    if dout.stride(3) != 1:
        dout = dout.contiguous()
    if packed_grad:
        buf = torch.empty((Bq, T, 3, H, hd), dtype=torch.bfloat16, device=q.device)
        dq, dk, dv = buf[:, :, 0], buf[:, :, 1], buf[:, :, 2]
    else:
        buf = None
        dq, dk, dv = (torch.empty((Bq, T, H, hd), dtype=torch.bfloat16, device=q.device) for _ in range(3))
    dsum = torch.empty((B, H, Tm), dtype=torch.float32, device=q.device)
    rc = _lib.lib().vaa_model_attention_bwd(q.data_ptr(), _str3(q), k.data_ptr(), _str3(k), v.data_ptr(), _str3(v), o.data_ptr(), _str3(o),
                                            dout.data_ptr(), _str3(dout), lse.data_ptr(), dsum.data_ptr(), dq.data_ptr(), _str3(dq),
                                            dk.data_ptr(), _str3(dk), dv.data_ptr(), _str3(dv), rope[0].data_ptr() if rope else None,
                                            rope[1].data_ptr() if rope else None, cu_seqlens.data_ptr() if cu_seqlens is not None else None,
                                            B, H, Tm, hd, int(bool(causal)), float(scale), _stream())
    _lib.check(rc, "vaa_model_attention_bwd")
    return buf if packed_grad else (dq, dk, dv)


class AttentionFn(torch.autograd.Function):
    """softmax(scale q k^T [causal]) v on [B,T,H,hd] bf16 views -> [B,T,H,hd] (vaa_attention.hip, fwd + 2-kernel bwd)."""

    @staticmethod
    def forward(ctx, q, k, v, causal, scale):
        scale = float(q.shape[-1]) ** -0.5 if scale is None else float(scale)
        o, lse = attention_fwd(q, k, v, causal, scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.causal, ctx.scale = bool(causal), scale
        return o

    @staticmethod
    def backward(ctx, dout):
        q, k, v, o, lse = ctx.saved_tensors
        dq, dk, dv = attention_bwd(q, k, v, o, lse, dout, ctx.causal, ctx.scale)
        return dq, dk, dv, None, None


class RopeAttentionFn(torch.autograd.Function):
    """rope(q), rope(k) -> causal attention, with the rotary adjoint of dq/dk fused into the backward kernels' epilogues
    (saves two full passes over [B,T,H,hd] per layer). q, k, v: [B,T,H,hd] bf16 views; cos/sin: float32 [T,hd/2]."""

    @staticmethod
    def forward(ctx, q, k, v, cos, sin, causal, scale, cu_seqlens=None, max_len=0):
        """Packed form (cu_seqlens given): q/k/v are [1,sumT,H,hd] and cos/sin hold one row PER PACKED TOKEN (gathered by position id)."""
        scale = float(q.shape[-1]) ** -0.5 if scale is None else float(scale)
        qr, kr = _rope_launch(q, cos, sin, 1.0), _rope_launch(k, cos, sin, 1.0)
        o, lse = attention_fwd(qr, kr, v, causal, scale, cu_seqlens, max_len)
        ctx.save_for_backward(qr, kr, v, o, lse, cos, sin, *([cu_seqlens] if cu_seqlens is not None else []))
        ctx.causal, ctx.scale, ctx.max_len = bool(causal), scale, int(max_len)
        return o

    @staticmethod
    def backward(ctx, dout):
        qr, kr, v, o, lse, cos, sin, *cu = ctx.saved_tensors
        dq, dk, dv = attention_bwd(qr, kr, v, o, lse, dout, ctx.causal, ctx.scale, rope=(cos, sin), cu_seqlens=cu[0] if cu else None,
                                   max_len=ctx.max_len)
        return dq, dk, dv, None, None, None, None, None, None


class RopePackedAttentionFn(torch.autograd.Function):
    """rope(q), rope(k) -> causal attention on a packed qkv [B,T,3,H,hd] — the output of ONE q/k/v GEMM over row-concatenated weights — with
    the gradient coming back as one [B,T,3,H,hd] buffer (dq | dk | dv slices, rotary adjoint applied in the backward kernels' epilogues): the
    layer's q/k/v data-gradient is then ONE GEMM with K = 3D as well. Same kernels as RopeAttentionFn, strided operands."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, causal, scale, cu_seqlens=None, max_len=0):
        scale = float(qkv.shape[-1]) ** -0.5 if scale is None else float(scale)
        qr, kr = _rope_launch(qkv[:, :, 0], cos, sin, 1.0), _rope_launch(qkv[:, :, 1], cos, sin, 1.0)
        v = qkv[:, :, 2]
        o, lse = attention_fwd(qr, kr, v, causal, scale, cu_seqlens, max_len)
        ctx.save_for_backward(qr, kr, qkv, o, lse, cos, sin, *([cu_seqlens] if cu_seqlens is not None else []))
        ctx.causal, ctx.scale, ctx.max_len = bool(causal), scale, int(max_len)
        return o

    @staticmethod
    def backward(ctx, dout):
        qr, kr, qkv, o, lse, cos, sin, *cu = ctx.saved_tensors
        buf = attention_bwd(qr, kr, qkv[:, :, 2], o, lse, dout, ctx.causal, ctx.scale, packed_grad=True, rope=(cos, sin),
                            cu_seqlens=cu[0] if cu else None, max_len=ctx.max_len)
        return buf, None, None, None, None, None, None


class PackedAttentionFn(torch.autograd.Function):
    """Same on a packed qkv [B,T,3,H,hd] (the ViT blocks' fused projection); the gradient comes back packed, in one buffer."""

    @staticmethod
    def forward(ctx, qkv, causal, scale):
        scale = float(qkv.shape[-1]) ** -0.5 if scale is None else float(scale)
        o, lse = attention_fwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal, scale)
        ctx.save_for_backward(qkv, o, lse)
        ctx.causal, ctx.scale = bool(causal), scale
        return o

    @staticmethod
    def backward(ctx, dout):
        qkv, o, lse = ctx.saved_tensors
        return attention_bwd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], o, lse, dout, ctx.causal, ctx.scale, packed_grad=True), None, None
