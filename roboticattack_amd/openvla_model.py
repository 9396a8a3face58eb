"""OpenVLA-7B-shaped model in plain PyTorch-ROCm (plumbing around the hot path, NOT a hand-written kernel target).

The reference keeps the model as stock PyTorch/HF/timm (prismatic/extern/hf/modeling_prismatic.py:63-447);
neither timm nor the OpenVLA weights exist in this image, so the repo carries the architecture itself:

  vision  : DINOv2 ViT-L/14-reg4 @224 (1024-d, 24 blocks, LayerScale, cls + 4 register tokens) and
            SigLIP so400m/14 @224 (1152-d, 27 blocks); each returns the patch tokens of its SECOND-TO-LAST block
            (`get_intermediate_layers(n={len(blocks)-2})`, modeling_prismatic.py:85-87, 99-101), concatenated on the
            feature axis -> [B,256,2176]; the 6-channel input is split 3+3 (modeling_prismatic.py:120).
  project : fused-backbone MLP 2176 -> 8704 -> 4096 -> 4096 with GELU (modeling_prismatic.py:139-156).
  llm     : Llama-2-7B (32 layers, 4096-d, 32 heads, SwiGLU 11008, RoPE, RMSNorm 1e-6... HF default 1e-5 for Llama-2),
            vocab padded to 32064 (configuration_prismatic.py:86); image tokens inserted after BOS (:383-385).

All weights are frozen (Appendix A-D12: freezing does not change the patch gradient), so autograd stores no GEMM
inputs and computes activation gradients only. Instead of materialising fp32 logits [B,S,32064] (2.4 GB at B=64) the
LM head runs on the labelled rows only (`forward_rows`), which is exactly the set of rows the reference's losses read.
Random-init throughput runs use `init="random"`; a local HF checkpoint directory can be mapped in with `load_hf_openvla`.
"""
from __future__ import annotations

import math
import types
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import IGNORE_INDEX, MODEL_VOCAB, N_IMG_TOKENS, PAD_ID


@dataclass
class VitCfg:
    dim: int
    depth: int
    heads: int
    mlp: int
    n_prefix: int  # cls + register tokens
    cls_pos: bool  # the class token has its own position-embedding slot (timm models WITHOUT no_embed_class); register tokens never do
    layerscale: bool


@dataclass
class OpenVLACfg:
    dino: VitCfg
    siglip: VitCfg
    llm_dim: int = 4096
    llm_layers: int = 32
    llm_heads: int = 32
    llm_mlp: int = 11008
    vocab: int = MODEL_VOCAB
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    # prompt-length bucket of the rows-only path: L is padded up to max(seq_floor, multiple of seq_multiple) so that the GEMM shapes of a
    # step do not depend on the longest prompt of the batch (the TunableOp selections then apply on every rank and every outer iteration;
    # 0 = off). 44 = the longest prompt of the BridgeData-shaped loaders: at bs=64 nearly every batch has that length anyway.
    seq_floor: int = 0
    seq_multiple: int = 1


def openvla_7b_cfg() -> OpenVLACfg:
    return OpenVLACfg(
        # timm vit_large_patch14_reg4_dinov2.lvd142m is a `no_embed_class` model: pos_embed is [1,256,1024] over the PATCH tokens only,
        # cls + 4 register tokens are concatenated afterwards (the checkpoint holds pos_embed 256, cls_token 1, reg_token 4)
        dino=VitCfg(1024, 24, 16, 4096, 5, False, True),
        siglip=VitCfg(1152, 27, 16, 4304, 0, False, False),
        seq_floor=44, seq_multiple=4,
    )


def tiny_cfg() -> OpenVLACfg:
    """Same topology, toy widths: CPU-runnable plumbing tests."""
    return OpenVLACfg(
        dino=VitCfg(64, 3, 2, 128, 5, False, True),
        siglip=VitCfg(128, 3, 2, 256, 0, False, False),  # tower widths are multiples of 64 so that the fused patch-embed backward applies on a GPU
        llm_dim=64, llm_layers=2, llm_heads=4, llm_mlp=128,
    )


class VitBlock(nn.Module):
    def __init__(self, c: VitCfg):
        super().__init__()
        self.heads = c.heads
        self.norm1 = nn.LayerNorm(c.dim, eps=1e-6)
        self.qkv = nn.Linear(c.dim, 3 * c.dim)
        self.proj = nn.Linear(c.dim, c.dim)
        self.norm2 = nn.LayerNorm(c.dim, eps=1e-6)
        self.fc1 = nn.Linear(c.dim, c.mlp)
        self.fc2 = nn.Linear(c.mlp, c.dim)
        self.ls1 = nn.Parameter(torch.ones(c.dim)) if c.layerscale else None
        self.ls2 = nn.Parameter(torch.ones(c.dim)) if c.layerscale else None

    def forward(self, x):
        B, T, D = x.shape
        from . import model_ops

        fused_ln = model_ops.enabled(x) and D % 8 == 0 and D <= 8192
        if fused_ln:  # (x) -> (x, LN(x)): the backward kernel also absorbs the residual-stream gradient
            x, h = model_ops.ResidualLayerNormFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        else:
            h = self.norm1(x)
        qkv = self.qkv(h).view(B, T, 3, self.heads, D // self.heads)
        if model_ops.attention_enabled(qkv):  # matrix-core attention on the packed projection; packed gradient, no permute copies
            a = model_ops.PackedAttentionFn.apply(qkv, False, None).reshape(B, T, D)
        else:
            qkv = qkv.permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(B, T, D)
        a = self.proj(a)
        x = model_ops.scale_add(x, a, self.ls1) if self.ls1 is not None else x + a  # LayerScale + residual in one pass
        if fused_ln:
            x, h = model_ops.ResidualLayerNormFn.apply(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        else:
            h = self.norm2(x)
        h = self.fc2(F.gelu(self.fc1(h)))
        return model_ops.scale_add(x, h, self.ls2) if self.ls2 is not None else x + h


class Vit(nn.Module):
    """timm VisionTransformer restricted to what `get_intermediate_layers(n={depth-2})` evaluates."""

    def __init__(self, c: VitCfg):
        super().__init__()
        self.c = c
        # timm's PatchEmbed is Conv2d(3, dim, 14, stride=14); kernel == stride, so it is exactly a GEMM over 588-pixel
        # tiles. Stored in conv layout (checkpoint compatible), evaluated as unfold + linear: hipBLASLt instead of
        # MIOpen (whose bf16 backward-data path for this shape goes through a slow find pass).
        self.patch_embed = nn.Conv2d(3, c.dim, kernel_size=14, stride=14)
        self.pos_embed = nn.Parameter(torch.zeros(1, N_IMG_TOKENS + (1 if c.cls_pos else 0), c.dim))
        self.prefix = nn.Parameter(torch.zeros(1, c.n_prefix, c.dim)) if c.n_prefix else None
        self.blocks = nn.ModuleList([VitBlock(c) for _ in range(c.depth - 1)])  # the last block is never evaluated

    def embed_params(self):
        """(W [D,588], bias [D], packed W^T, cached) of the patch-embed GEMM — what ops.PatchApplyEmbed needs. The packed weights are
        ops.pack_embed_weights(W^T [588,D]): K2''s MFMA fragment order, rebuilt only when the parameter changes."""
        from . import ops
        w = self.patch_embed.weight
        if not (w.is_cuda and w.dtype == torch.bfloat16 and self.c.dim % 64 == 0):  # no fused backward for this tower: nothing to pack
            return w.detach().reshape(self.c.dim, 588), self.patch_embed.bias.detach(), None
        hit = getattr(self, "_embed_wp", None)
        if hit is None or hit[0] != w._version or hit[1] is not w:
            hit = (w._version, w, ops.pack_embed_weights(w.detach().reshape(self.c.dim, 588).t().contiguous()))
            self._embed_wp = hit
        return w.detach().reshape(self.c.dim, 588), self.patch_embed.bias.detach(), hit[2]

    def forward(self, img, embedded=None):
        """`embedded` [B,256,D]: the patch-embed output computed elsewhere (ops.PatchApplyEmbed) — `img` is then ignored."""
        if embedded is not None:
            x = embedded
        else:
            B0 = img.shape[0]
            tiles = img.reshape(B0, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(B0, 256, 588)
            x = F.linear(tiles, self.patch_embed.weight.reshape(self.c.dim, 588), self.patch_embed.bias)  # [B,256,D]
        if self.prefix is not None and self.c.cls_pos:
            B = x.shape[0]
            cls = self.prefix[:, :1].expand(B, -1, -1)
            reg = self.prefix[:, 1:].expand(B, -1, -1)
            x = torch.cat([cls, x], dim=1) + self.pos_embed  # timm without no_embed_class: pos-embed over [cls, patches], then registers
            x = torch.cat([x[:, :1], reg, x[:, 1:]], dim=1)
        elif self.prefix is not None:
            # timm `no_embed_class` (DINOv2 reg4, VisionTransformer._pos_embed): position embedding on the patch tokens only, then
            # [cls, registers] are prepended
            x = torch.cat([self.prefix.expand(x.shape[0], -1, -1), x + self.pos_embed], dim=1)
        else:
            x = x + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return x[:, self.c.n_prefix :]


class RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        return F.rms_norm(x, (x.shape[-1],), self.weight, self.eps)


def _rope(q, k, cos, sin):
    def rot(x):
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2 :]
        return torch.cat((-x2, x1), dim=-1)

    return q * cos + rot(q) * sin, k * cos + rot(k) * sin


class LlamaLayer(nn.Module):
    def __init__(self, c: OpenVLACfg):
        super().__init__()
        d = c.llm_dim
        self.heads = c.llm_heads
        self.input_layernorm = RMSNorm(d, c.rms_eps)
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.o_proj = nn.Linear(d, d, bias=False)
        self.post_attention_layernorm = RMSNorm(d, c.rms_eps)
        self.gate_proj = nn.Linear(d, c.llm_mlp, bias=False)
        self.up_proj = nn.Linear(d, c.llm_mlp, bias=False)
        self.down_proj = nn.Linear(c.llm_mlp, d, bias=False)
        self._wt_cache = {}

    def _wt(self, name):
        """Resident transposed copy of a frozen projection ([in,out] contiguous) for the TN-layout dgrad (model_ops.FrozenLinearsFn)."""
        w = getattr(self, name).weight
        hit = self._wt_cache.get(name)
        if hit is None or hit[0] != w._version or hit[1] is not w:
            hit = (w._version, w, w.detach().t().contiguous())
            self._wt_cache[name] = hit
        return hit[2]

    def _wcat(self, names):
        """Row-concatenated frozen weights of several projections that share their input ([sum out, in]) and the transposed copy, built once."""
        ws = [getattr(self, n).weight for n in names]
        key = "+".join(names)
        ver = tuple(w._version for w in ws)
        hit = self._wt_cache.get(key)
        if hit is None or hit[0] != ver or any(a is not b for a, b in zip(hit[1], ws)):
            w = torch.cat([w.detach() for w in ws], 0).contiguous()
            hit = (ver, ws, w, w.t().contiguous())
            self._wt_cache[key] = hit
        return hit[2], hit[3]

    @staticmethod
    def _fuse_qkv(rows: int) -> bool:
        """q/k/v as ONE GEMM over the concatenated weights (and one data-gradient GEMM with K = 3D): at the per-rank batches of the
        strong-scaling configs (M = B*T = 1,200 / 2,400 rows) three [M,4096]x[4096,4096] products run at 0.8-1.05 PFLOP/s, the single
        [M,4096]x[4096,12288] at 1.3-1.36 (tools/gemm_overlap_probe.py); at M = 19,200 the two forms are level, so the separate, tuned GEMMs
        stay. VAA_FUSED_QKV=1 / 0 forces it on / off."""
        import os

        mode = os.environ.get("VAA_FUSED_QKV", "auto")
        return mode == "1" or (mode != "0" and rows < 8192)

    def _lin(self, x, names, res=None):
        from . import model_ops

        ws = []
        for n in names:
            ws += [getattr(self, n).weight, self._wt(n)]
        return model_ops.FrozenLinearsFn.apply(x, res, *ws)

    def forward(self, x, cos, sin, rope_tab=None, rows=None, pack=None):
        """`rows` (flat indices into the B*T positions): everything after the attention — o_proj, residual, MLP — is evaluated
        for those rows only and [R,D] is returned (last layer of `forward_rows`: no other position reaches the loss)."""
        from . import model_ops

        B, T, D = x.shape
        hd = D // self.heads
        fused = rope_tab is not None and model_ops.enabled(x) and hd % 16 == 0 and D % 8 == 0 and D <= 8192
        tn = fused and model_ops.tn_dgrad_enabled()
        if fused:
            x, h = model_ops.ResidualRMSNormFn.apply(x, self.input_layernorm.weight, self.input_layernorm.eps)
        else:
            h = self.input_layernorm(x)
        if tn and hd in (64, 128) and self._fuse_qkv(B * T):
            wqkv, wqkv_t = self._wcat(("q_proj", "k_proj", "v_proj"))
            (qkv,) = model_ops.FrozenLinearsFn.apply(h, None, wqkv, wqkv_t)
            qkv = qkv.view(B, T, 3, self.heads, hd)
            if model_ops.attention_enabled(qkv[:, :, 0]):
                a = model_ops.RopePackedAttentionFn.apply(qkv, *rope_tab, True, None, pack.cu if pack is not None else None,
                                                          pack.max_len if pack is not None else 0).view(B, T, D)
                q = k = v = None
            else:
                q, k, v = (qkv[:, :, z].reshape(B, T, D) for z in range(3))
                a = None
        elif tn:
            q, k, v = self._lin(h, ("q_proj", "k_proj", "v_proj"))
            a = None
        else:
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            a = None
        if a is not None:
            pass
        elif fused and model_ops.attention_enabled(q.view(B, T, self.heads, hd)):
            sh = (B, T, self.heads, hd)
            if hd in (64, 128):  # rotary adjoint of dq/dk runs in the attention backward's epilogues
                a = model_ops.RopeAttentionFn.apply(q.view(sh), k.view(sh), v.view(sh), *rope_tab, True, None,
                                                    pack.cu if pack is not None else None, pack.max_len if pack is not None else 0).view(B, T, D)
            else:
                q = model_ops.RopeFn.apply(q.view(sh), *rope_tab)
                k = model_ops.RopeFn.apply(k.view(sh), *rope_tab)
                a = model_ops.AttentionFn.apply(q, k, v.view(sh), True, None).view(B, T, D)
        elif fused:  # one HBM pass per tensor instead of neg + cat + 2 mul + add (and their autograd chains)
            q = model_ops.RopeFn.apply(q.view(B, T, self.heads, hd), *rope_tab).transpose(1, 2)
            k = model_ops.RopeFn.apply(k.view(B, T, self.heads, hd), *rope_tab).transpose(1, 2)
            v = v.view(B, T, self.heads, hd).transpose(1, 2)
            a = None
        else:
            q = q.view(B, T, self.heads, hd).transpose(1, 2)
            k = k.view(B, T, self.heads, hd).transpose(1, 2)
            v = v.view(B, T, self.heads, hd).transpose(1, 2)
            q, k = _rope(q, k, cos, sin)
            a = None
        if a is None:
            a = F.scaled_dot_product_attention(q, k, v, is_causal=True)  # right padding + causal == HF's mask on real tokens
            a = a.transpose(1, 2).reshape(B, T, D)
        if rows is not None:
            a = a.reshape(-1, D).index_select(0, rows)[None]
            x = x.reshape(-1, D).index_select(0, rows)[None]
            B, T = 1, a.shape[1]
        if tn:
            (x,) = self._lin(a, ("o_proj",), res=x)  # residual in the GEMM epilogue
        else:
            x = torch.addmm(x.reshape(-1, D), a.reshape(-1, D), self.o_proj.weight.t()).view(B, T, D)
        if fused:
            x, h = model_ops.ResidualRMSNormFn.apply(x, self.post_attention_layernorm.weight, self.post_attention_layernorm.eps)
        else:
            h = self.post_attention_layernorm(x)
        if fused and (h.shape[0] * h.shape[1] * self.gate_proj.out_features) % 8 == 0:
            if tn:
                g, u = self._lin(h, ("gate_proj", "up_proj"))
                (x,) = self._lin(model_ops.SwiGLUFn.apply(g, u), ("down_proj",), res=x)
                return x[0] if rows is not None else x
            y = model_ops.SwiGLUFn.apply(self.gate_proj(h), self.up_proj(h))
            x = torch.addmm(x.reshape(-1, D), y.reshape(-1, y.shape[-1]), self.down_proj.weight.t()).view(B, T, D)
        else:
            x = x + self.down_proj(F.silu(self.gate_proj(h)) * self.up_proj(h))
        return x[0] if rows is not None else x


class SeqPack:
    """Right-padded batch -> packed token axis for the Llama stack: the padding rows (8 of 300 tokens per sample on BridgeData-like
    prompts) are never computed. Built once per outer iteration from `attention_mask` (one host sync for the lengths); the multimodal
    sequence of sample b is [BOS, 256 image tokens, text 1..len_b-1], i.e. 256 + len_b tokens of the padded 256 + L."""

    def __init__(self, attention_mask: torch.Tensor):
        B, L = attention_mask.shape
        self.T_pad = N_IMG_TOKENS + L
        lens = (attention_mask.sum(1).to(torch.int64) + N_IMG_TOKENS).tolist()
        dev = attention_mask.device
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + int(n))
        self.total = cu[-1]  # real tokens
        fill = (-cu[-1]) % 256  # GEMM row count kept a multiple of the 256-row MFMA macro tile: a dummy sequence of `fill` tokens
        if fill:                # (copies of token 0, attended only by itself, never selected, zero gradient)
            lens = lens + [fill]
            cu.append(cu[-1] + fill)
        self.lens, self.max_len = lens, max(lens)
        self.cu = torch.tensor(cu, dtype=torch.int32, device=dev)
        self.pos = torch.cat([torch.arange(n, device=dev) for n in lens])                                   # position id of every packed token
        self.gather = torch.cat([torch.arange(n, device=dev) + b * self.T_pad if b < B else torch.zeros(n, dtype=torch.int64, device=dev)
                                 for b, n in enumerate(lens)])  # packed <- padded flat index
        self._rope = None

    def rows(self, row_index_padded: torch.Tensor) -> torch.Tensor:
        """Padded flat indices b*T_pad + t -> packed indices cu[b] + t."""
        b = torch.div(row_index_padded, self.T_pad, rounding_mode="floor")
        return self.cu.to(torch.int64).index_select(0, b) + (row_index_padded - b * self.T_pad)

    def rope_tab(self, cos_half: torch.Tensor, sin_half: torch.Tensor):
        """One rotary row per packed token (gathered by position id), cached."""
        if self._rope is None:
            self._rope = (cos_half.index_select(0, self.pos).contiguous(), sin_half.index_select(0, self.pos).contiguous())
        return self._rope


class OpenVLAShaped(nn.Module):
    def __init__(self, cfg: OpenVLACfg | None = None):
        super().__init__()
        self.cfg = c = cfg or openvla_7b_cfg()
        self.featurizer = Vit(c.dino)
        self.fused_featurizer = Vit(c.siglip)
        vd = c.dino.dim + c.siglip.dim
        self.fc1 = nn.Linear(vd, 4 * vd)
        self.fc2 = nn.Linear(4 * vd, c.llm_dim)
        self.fc3 = nn.Linear(c.llm_dim, c.llm_dim)
        self.embed_tokens = nn.Embedding(c.vocab, c.llm_dim)
        self.layers = nn.ModuleList([LlamaLayer(c) for _ in range(c.llm_layers)])
        self.norm = RMSNorm(c.llm_dim, c.rms_eps)
        self.lm_head = nn.Linear(c.llm_dim, c.vocab, bias=False)
        # the attack loops read vla.vision_backbone.featurizer.patch_embed.num_patches (UADA.py:166)
        self.vision_backbone = types.SimpleNamespace(
            featurizer=types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=N_IMG_TOKENS)))
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def device(self):
        return self.embed_tokens.weight.device

    @torch.no_grad()
    def init_random(self, seed: int = 0, std: float = 0.02):
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() >= 2:
                p.normal_(0.0, std, generator=g)
            elif name.endswith("bias"):
                p.zero_()
            elif "ls1" in name or "ls2" in name:
                p.fill_(1.0)
        return self

    def patch_embed_params(self):
        """(w0, b0, wp0, w1, b1, wp1) for ops.PatchApplyEmbed, or None when the fused patch-embed backward does not apply
        (tower widths must be multiples of 64, bf16 on a ROCm device)."""
        w = self.featurizer.patch_embed.weight
        if not w.is_cuda or w.dtype != torch.bfloat16 or self.cfg.dino.dim % 64 or self.cfg.siglip.dim % 64:
            return None
        return (*self.featurizer.embed_params(), *self.fused_featurizer.embed_params())

    def _towers(self, in0, in1, embedded: bool):
        """The two vision towers are independent until the projector. At the small per-rank batches of the strong-scaling configs
        (bs = 8 / 4: ViT GEMMs of ~2,000 rows, 36-130 output tiles for 256 CUs) one tower leaves most of the chip idle, so the SigLIP
        tower runs on a second HIP stream next to DINOv2 — forward, and backward too (autograd replays each node on its forward
        stream). Measured on one MI355X: bs=8 78.2 -> 74.7 ms/step, bs=64 473 -> 468 (the towers' tails overlap). VAA_TOWER_STREAMS=0
        turns it off."""
        import os

        f = (lambda m, x: m(None, embedded=x)) if embedded else (lambda m, x: m(x))
        mode = os.environ.get("VAA_TOWER_STREAMS", "auto")
        if mode == "auto" and os.environ.get("VAA_DIST_BACKEND") == "gloo" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            mode = "0"  # test mode, several ranks share one GPU: their four queues time-slice badly (strong region 609 -> 2,270+ ms per step measured)
        if not in0.is_cuda or mode == "0":
            return f(self.featurizer, in0), f(self.fused_featurizer, in1)
        cur = torch.cuda.current_stream(in0.device)
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != in0.device:
            side = self._side_stream = torch.cuda.Stream(device=in0.device)
        side.wait_stream(cur)
        in1.record_stream(side)  # allocated on the main stream, read on the side stream
        with torch.cuda.stream(side):
            f1 = f(self.fused_featurizer, in1)
        f0 = f(self.featurizer, in0)
        cur.wait_stream(side)
        f1.record_stream(cur)
        return f0, f1

    def _rope_tables(self, T: int, device, dtype):
        """(cos, sin, (cos_half, sin_half)) of the T positions of a prompt bucket, built ONCE per (T, device, dtype): they are constants of the
        model, and rebuilding them every step cost ~10 small launches (arange / pow / reciprocal / outer / cat / cos / sin / casts, ~55 us —
        three quarters of the whole hand-written path). Same ops as before on first use: bitwise the per-step tables."""
        key = (int(T), str(device), dtype)
        cache = self.__dict__.setdefault("_rope_cache", {})
        hit = cache.get(key)
        if hit is None:
            hd = self.cfg.llm_dim // self.cfg.llm_heads
            with torch.no_grad():
                inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, device=device, dtype=torch.float32) / hd))
                ang = torch.outer(torch.arange(T, device=device, dtype=torch.float32), inv)
                ang = torch.cat([ang, ang], dim=-1)
                cos, sin = ang.cos().to(dtype)[None, None], ang.sin().to(dtype)[None, None]
                half = ang[:, : hd // 2]
                # fused path: HF rounds cos/sin to the activation dtype before use (modeling_llama rotary) — keep that rounding
                tab = (half.cos().to(dtype).float().contiguous(), half.sin().to(dtype).float().contiguous())
            if len(cache) >= 16:  # prompt buckets are few; never grow without bound
                cache.clear()
            hit = cache[key] = (cos, sin, tab)
        return hit

    def seq_bucket(self, L: int) -> int:
        """Padded prompt length of the rows-only path (cfg.seq_floor / seq_multiple; VAA_SEQ_FLOOR overrides the floor, 0 = off)."""
        import os

        floor = int(os.environ.get("VAA_SEQ_FLOOR", self.cfg.seq_floor))
        if floor <= 0 or os.environ.get("VAA_SEQ_PACK"):
            return L
        m = max(1, self.cfg.seq_multiple)
        return max(floor, (L + m - 1) // m * m)

    def hidden_states(self, input_ids, pixel_values, rows=None, patch_embeds=None, pack=None):
        """[B, 1+256+(L-1), D] final-norm hidden states of the multimodal sequence (modeling_prismatic.py:366-415); with
        `rows` (flat position indices into the PADDED [B*T] layout) only those positions of the LAST layer are evaluated and
        [R,D] is returned. `pack` (SeqPack): the Llama stack runs on the packed token axis ([1, sum T_b, D] is returned unless
        `rows` is given)."""
        if patch_embeds is not None:
            in0, in1, kw = patch_embeds[0], patch_embeds[1], True
        else:
            (in0, in1), kw = torch.split(pixel_values, [3, 3], dim=1), False
        f0, f1 = self._towers(in0, in1, kw)
        feats = torch.cat([f0, f1], dim=2)
        proj = self.fc3(F.gelu(self.fc2(F.gelu(self.fc1(feats)))))
        if rows is not None and pack is None:
            # rows-only path: right-pad the prompts to the bucket length (causal attention: the labelled rows do not see the extra pad
            # positions); `rows` from label_row_index already addresses the bucketed layout
            # the padded length travels WITH the index (label_row_index tags it): the two sides cannot disagree when the environment
            # changes between the calls; an untagged index (built by other means for the documented [B*(256+L)] layout) must address
            # the bucketed layout of seq_bucket(L) itself
            S_rows = getattr(rows, "vaa_S", None)
            Lb = (int(S_rows) - N_IMG_TOKENS) if S_rows is not None else self.seq_bucket(input_ids.shape[1])
            if Lb < input_ids.shape[1]:
                raise ValueError(f"hidden_states: the row index was built for prompts of {Lb} tokens, input_ids has {input_ids.shape[1]}")
            if Lb != input_ids.shape[1]:
                input_ids = F.pad(input_ids, (0, Lb - input_ids.shape[1]), value=PAD_ID)
        emb = self.embed_tokens(input_ids)
        x = torch.cat([emb[:, :1], proj.to(emb.dtype), emb[:, 1:]], dim=1)
        T = x.shape[1]
        hd = self.cfg.llm_dim // self.cfg.llm_heads
        cos, sin, rope_tab = self._rope_tables(T, x.device, x.dtype)
        from . import model_ops

        if pack is not None and model_ops.enabled(x) and model_ops.attention_enabled(x.view(x.shape[0], T, self.cfg.llm_heads, hd)) and hd in (64, 128):
            # packed token axis [1, sum T_b, D]: no padding rows through the 32 layers (rows = packed indices, see SeqPack.rows)
            x = x.reshape(-1, x.shape[-1]).index_select(0, pack.gather)[None]
            rope_tab = pack.rope_tab(*rope_tab)
            rows = pack.rows(rows) if rows is not None else None
        else:
            pack = None
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            x = layer(x, cos, sin, rope_tab, rows if i == last else None, pack)
        return self.norm(x)

    @staticmethod
    def make_pack(attention_mask):
        """SeqPack of a right-padded batch when packing is asked for (VAA_SEQ_PACK=1) and there is something to drop, else None.
        OPT-IN: the packed row count (e.g. 18 688 instead of 19 200 at bs=64) falls outside the shipped hipBLASLt selections, and the
        default heuristics cost more (476 -> 511 ms/step measured) than the 2.7 % of rows saved; it pays once those shapes are tuned."""
        import os

        if not os.environ.get("VAA_SEQ_PACK") or attention_mask is None or not attention_mask.is_cuda:
            return None
        p = SeqPack(attention_mask)
        return p if p.total < attention_mask.shape[0] * p.T_pad else None

    def label_row_index(self, labels):
        """Flat indices (into the [B*S] hidden rows, S = 256 + seq_bucket(L)) of the labelled rows in (b,k) row-major order of
        labels[b,k+1] != -100. Costs one host sync (nonzero); the attack loops compute it once per outer iteration."""
        B, L = labels.shape
        S = N_IMG_TOKENS + self.seq_bucket(L)
        bk = (labels[:, 1:] != IGNORE_INDEX).nonzero(as_tuple=False)  # [R,2] sorted row-major
        idx = bk[:, 0] * S + N_IMG_TOKENS + bk[:, 1]
        idx.vaa_S = S  # hidden_states pads the prompts to THIS length (the environment is not consulted a second time)
        return idx

    def forward_rows(self, input_ids, pixel_values, labels, row_index=None, patch_embeds=None, pack=None):
        """Logits [R,V] of the labelled rows only, in (b,k) row-major order of labels[b,k+1] != -100 (VAA_LAYOUT_ROWS):
        row (b,k) is model position S-L+k = 256+k, the position whose next-token target is labels[b,k+1].
        Pass `row_index=label_row_index(labels)` to keep the step free of host synchronisation."""
        if row_index is None:
            row_index = self.label_row_index(labels)
        return self.lm_head(self.hidden_states(input_ids, pixel_values, rows=row_index, patch_embeds=patch_embeds, pack=pack))  # [R, D] -> [R, V]

    def hidden_rows(self, input_ids, pixel_values, row_index, patch_embeds=None, pack=None):
        """Final-norm hidden states [R,D] of the rows `row_index` (label_row_index): the input of the LM head. `ops.HeadLossRows` applies the
        head and the loss to them so that the head's backward can contract over the action columns only."""
        return self.hidden_states(input_ids, pixel_values, rows=row_index, patch_embeds=patch_embeds, pack=pack)

    def forward(self, input_ids, attention_mask=None, pixel_values=None, labels=None, **_):
        """Drop-in contract of PrismaticForConditionalGeneration.forward: full fp32 logits and HF's mean CE."""
        from .surrogate import hf_causal_ce

        logits = self.lm_head(self.hidden_states(input_ids, pixel_values)).float()
        loss = hf_causal_ce(logits, labels) if labels is not None else None
        return types.SimpleNamespace(loss=loss, logits=logits)


def enable_tuned_gemms() -> bool:
    """Point PyTorch-ROCm's TunableOp at the GEMM selections recorded on an MI355X for the OpenVLA-7B step at the per-rank batches of the
    BASELINE configs — bs = 64 (weak scaling), 32 / 16 / 8 (strong scaling of config 3 over 2 / 4 / 8 ranks; config 4: 32 over 4 ranks) and
    4 (config 5) with prompts bucketed to 44 tokens (cfg.seq_floor) — (roboticattack_amd/tunableop/*.csv, one identical copy per device ordinal; made by
    tools/tune_gemms.sh). Tuning itself stays off: unknown shapes fall back to the default hipBLASLt heuristic, a validator mismatch
    (different ROCm / hipBLASLt build) ignores the file. Worth ~3 % of the step; honours a user's own PYTORCH_TUNABLEOP_* env."""
    import os

    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None or os.environ.get("VAA_NO_TUNED_GEMMS"):
        return False
    try:
        import torch.cuda.tunable as tunable

        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop", "openvla7b_mi355x.csv"), insert_device_ordinal=True)
        return True
    except Exception:
        return False


def build_openvla(cfg: OpenVLACfg | None = None, device="cuda", dtype=torch.bfloat16, seed: int = 0) -> OpenVLAShaped:
    """Random-init OpenVLA-7B-shaped model created directly on `device` in `dtype` (15 GB in bf16)."""
    if torch.device(device).type == "cuda" and dtype == torch.bfloat16:
        enable_tuned_gemms()
    # parameters are created in `dtype` directly (no fp32 copy of the 7.5 B parameters: 30 GB of transient HBM per process, which 8 ranks
    # sharing one GPU in the functional multi-rank runs cannot afford); init_random overwrites every matrix, the 1-D defaults (ones / zeros)
    # are exact in any dtype
    prev = torch.get_default_dtype()
    try:
        if dtype.is_floating_point:
            torch.set_default_dtype(dtype)
        with torch.device(device):
            m = OpenVLAShaped(cfg)
    finally:
        torch.set_default_dtype(prev)
    m = m.to(dtype)
    m.init_random(seed)
    return m.eval()


def check_hf_config(cfg: OpenVLACfg, ckpt_dir: str) -> list:
    """The checkpoint's `config.json` (OpenVLAConfig: configuration_prismatic.py:83-123; `text_config` = the Llama config, :119-123) against the
    constants this module tree is built with. Returns the list of disagreements (empty = the checkpoint fits): a checkpoint whose shapes or
    numerical constants differ — RMSNorm epsilon, RoPE base, vocabulary / padding, image size, a single vision tower — must not be loaded
    into a model that silently computes with other values. A checkpoint directory without config.json is refused as well."""
    import json
    import os

    path = os.path.join(ckpt_dir, "config.json")
    if not os.path.exists(path):
        return [f"no config.json in {ckpt_dir}"]
    with open(path) as f:
        conf = json.load(f)
    tc = conf.get("text_config") or {}
    bad = []

    def need(what, got, want, tol=0.0):
        if got is None:
            return
        ok = abs(float(got) - float(want)) <= tol * abs(float(want)) if isinstance(want, float) else got == want
        if not ok:
            bad.append(f"{what} = {got!r}, model has {want!r}")

    need("text_config.hidden_size", tc.get("hidden_size"), cfg.llm_dim)
    need("text_config.num_hidden_layers", tc.get("num_hidden_layers"), cfg.llm_layers)
    need("text_config.num_attention_heads", tc.get("num_attention_heads"), cfg.llm_heads)
    need("text_config.num_key_value_heads", tc.get("num_key_value_heads"), cfg.llm_heads)  # no grouped-query attention in this module tree
    need("text_config.intermediate_size", tc.get("intermediate_size"), cfg.llm_mlp)
    need("text_config.vocab_size", tc.get("vocab_size"), cfg.vocab)
    # transformers' LlamaConfig defaults when a key is absent from text_config: rms_norm_eps 1e-6, rope_theta 10000.0
    need("text_config.rms_norm_eps", float(tc.get("rms_norm_eps", 1e-6)), float(cfg.rms_eps), 1e-9)
    need("text_config.rope_theta", float(tc.get("rope_theta", 10000.0)), float(cfg.rope_theta), 1e-9)
    if tc.get("rope_scaling") not in (None, {}):
        bad.append(f"text_config.rope_scaling = {tc.get('rope_scaling')!r}: not supported")
    need("pad_token_id", conf.get("pad_token_id"), 32000)          # configuration_prismatic.py:101; the loops pad with it (UADA.py:48)
    need("text_config.pad_token_id", tc.get("pad_token_id"), 32000)
    sizes = conf.get("image_sizes")
    if sizes is not None and [int(v) for v in sizes] != [224, 224]:
        bad.append(f"image_sizes = {sizes!r}, the patch transform and K1 are built for 224 x 224 frames")
    if conf.get("use_fused_vision_backbone") is False:
        bad.append("use_fused_vision_backbone = false: this module tree has the two fused towers (DINOv2 + SigLIP)")
    return bad


def load_hf_openvla(model: OpenVLAShaped, ckpt_dir: str) -> OpenVLAShaped:
    """Map a local HF `openvla/openvla-7b` safetensors checkpoint onto this module tree (names follow
    modeling_prismatic.py: vision_backbone.{featurizer,fused_featurizer}.*, projector.fc{1,2,3}, language_model.model.*).
    `config.json` is read first (check_hf_config): a checkpoint whose shapes or constants (rms_norm_eps, rope_theta, pad_token_id, image size)
    disagree with the model's OpenVLACfg is refused. The name mapping is exercised on a synthetic checkpoint written with the HF module names
    (tests/test_host_logic.py); the released weights themselves are not in this image (no network), so no numerical check against them exists."""
    import glob
    import os

    from safetensors.torch import load_file

    bad = check_hf_config(model.cfg, ckpt_dir)
    if bad:
        raise ValueError(f"load_hf_openvla: {ckpt_dir}/config.json disagrees with the model this loader builds ({'; '.join(bad)}) — build the "
                         "model with an OpenVLACfg that carries the checkpoint's values")
    sd = {}
    for f in sorted(glob.glob(os.path.join(ckpt_dir, "*.safetensors"))):
        sd.update(load_file(f))
    own = dict(model.named_parameters())
    ren = {}
    for k, v in sd.items():
        n = k
        n = n.replace("vision_backbone.featurizer.", "featurizer.").replace("vision_backbone.fused_featurizer.", "fused_featurizer.")
        n = n.replace("projector.", "").replace("language_model.model.", "").replace("language_model.lm_head.", "lm_head.")
        n = n.replace(".attn.qkv.", ".qkv.").replace(".attn.proj.", ".proj.").replace(".mlp.fc1.", ".fc1.").replace(".mlp.fc2.", ".fc2.")
        n = n.replace(".ls1.scale_factor", ".ls1").replace(".ls2.scale_factor", ".ls2").replace(".ls1.gamma", ".ls1").replace(".ls2.gamma", ".ls2")
        n = n.replace("patch_embed.proj.", "patch_embed.").replace(".self_attn.", ".").replace(".mlp.", ".")
        ren[n] = v
    missing = []
    with torch.no_grad():
        for name, p in own.items():
            if name == "featurizer.prefix":
                p.copy_(torch.cat([ren["featurizer.cls_token"], ren["featurizer.reg_token"]], dim=1))
            elif name in ren and ren[name].shape == p.shape:
                p.copy_(ren[name])
            else:
                missing.append(name)
    if missing:
        raise RuntimeError(f"load_hf_openvla: {len(missing)} parameters not found, e.g. {missing[:5]}")
    return model
