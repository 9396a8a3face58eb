"""Optional fused elementwise operators for the PyTorch-ROCm model (include/vaa_model_ops.h): RoPE and SwiGLU.

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
