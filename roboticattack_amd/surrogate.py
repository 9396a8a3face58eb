"""Tiny deterministic fp32 surrogate with OpenVLA's forward contract (plumbing / trajectory parity only).

Parity of optimised patches cannot be asserted through a bf16 7B model (GEMM rounding is amplified by
Adam's normalisation), so multi-step trajectories are compared on this surrogate: same
inputs/outputs as `PrismaticForConditionalGeneration.forward` (modeling_prismatic.py:291-447):

    forward(input_ids[B,L], attention_mask[B,L], pixel_values[B,6,224,224], labels[B,L]) ->
        .loss   mean CE over shifted non-ignored labels (HF Llama semantics, 256 image slots labelled -100)
        .logits f32 [B, 1+256+(L-1), 32064]

Weights are drawn on CPU from a seeded generator so that CPU and GPU instances are bit-identical.
"""
from __future__ import annotations

import math
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from .constants import IGNORE_INDEX, MODEL_VOCAB, N_IMG_TOKENS


def hf_causal_ce(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """HF LlamaForCausalLM loss on the multimodal sequence [3p transformers 4.40.1 modeling_llama.py]:
    labels get 256 x (-100) after BOS (modeling_prismatic.py:395-401), then shift-by-one mean CE in fp32."""
    B = labels.shape[0]
    mm = torch.cat(
        [labels[:, :1], torch.full((B, N_IMG_TOKENS), IGNORE_INDEX, dtype=labels.dtype, device=labels.device), labels[:, 1:]],
        dim=1,
    )
    shift_logits = logits[:, :-1, :].float()
    shift_labels = mm[:, 1:]
    return F.cross_entropy(
        shift_logits.reshape(-1, shift_logits.shape[-1]), shift_labels.reshape(-1), ignore_index=IGNORE_INDEX
    )


class SurrogateVLA(nn.Module):
    def __init__(self, d: int = 48, vocab: int = MODEL_VOCAB, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.patch_w = nn.Parameter(torch.randn(d, 6, 14, 14, generator=g) * 0.02, requires_grad=False)
        self.tok = nn.Parameter(torch.randn(vocab, d, generator=g) * 0.5, requires_grad=False)
        self.mix = nn.Parameter(torch.randn(d, d, generator=g) / math.sqrt(d), requires_grad=False)
        self.head = nn.Parameter(torch.randn(d, vocab, generator=g) * 0.3, requires_grad=False)
        # the attack loops read vla.vision_backbone.featurizer.patch_embed.num_patches (UADA.py:166)
        self.vision_backbone = types.SimpleNamespace(
            featurizer=types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=N_IMG_TOKENS))
        )

    @property
    def device(self) -> torch.device:
        return self.patch_w.device

    def forward(self, input_ids, attention_mask=None, pixel_values=None, labels=None, **_):
        x = F.conv2d(pixel_values.float(), self.patch_w, stride=14)  # [B,d,16,16]
        x = x.flatten(2).transpose(1, 2)  # [B,256,d]
        te = self.tok[input_ids]  # [B,L,d]
        h = torch.cat([te[:, :1], x, te[:, 1:]], dim=1)  # image tokens inserted after BOS
        S = h.shape[1]
        denom = torch.arange(1, S + 1, device=h.device, dtype=h.dtype)[None, :, None]
        c = torch.cumsum(h, dim=1) / denom  # causal prefix mean (position s sees tokens <= s)
        h2 = torch.tanh(c @ self.mix) + 0.25 * h
        logits = h2 @ self.head
        loss = hf_causal_ce(logits, labels) if labels is not None else None
        return types.SimpleNamespace(loss=loss, logits=logits)


class SurrogateEmbedVLA(nn.Module):
    """SurrogateVLA whose 256 image tokens come from TWO bf16 patch-embed towers — timm's PatchEmbed (Conv2d(3, D, 14, stride 14),
    modeling_prismatic.py:120-123) evaluated as a GEMM over 14x14 tiles — followed by an fp32 projector: the smallest model on which the
    production backward K2' (patch-embed backward on the tiles under the patch + gather) runs, so that K2' and the fused
    update can be put on a REFERENCE-LOOP trajectory (tools/gen_golden.py:gen_trajectory_k2e drives the reference's own UADA loop over this
    very module on the CPU; tests/test_gpu_attack.py replays it through K1 tile-major -> K2' -> epilogue + K4).

    It offers both boundaries: `forward(pixel_values=...)` (what the reference calls; bf16 pixel_values as UADA.py:142 casts them) and the
    rows / patch-embed interface of OpenVLAShaped (`patch_embed_params`, `label_row_index`, `forward_rows(..., patch_embeds=)`)."""

    def __init__(self, d: int = 48, D0: int = 64, D1: int = 128, vocab: int = MODEL_VOCAB, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        bf = torch.bfloat16
        self.w0 = nn.Parameter((torch.randn(D0, 588, generator=g) * 0.02).to(bf), requires_grad=False)
        self.b0 = nn.Parameter((torch.randn(D0, generator=g) * 0.01).to(bf), requires_grad=False)
        self.w1 = nn.Parameter((torch.randn(D1, 588, generator=g) * 0.02).to(bf), requires_grad=False)
        self.b1 = nn.Parameter((torch.randn(D1, generator=g) * 0.01).to(bf), requires_grad=False)
        self.proj = nn.Parameter(torch.randn(D0 + D1, d, generator=g) / math.sqrt(D0 + D1), requires_grad=False)
        self.tok = nn.Parameter(torch.randn(vocab, d, generator=g) * 0.5, requires_grad=False)
        self.mix = nn.Parameter(torch.randn(d, d, generator=g) / math.sqrt(d), requires_grad=False)
        self.head = nn.Parameter(torch.randn(d, vocab, generator=g) * 0.3, requires_grad=False)
        self.vision_backbone = types.SimpleNamespace(
            featurizer=types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=N_IMG_TOKENS))
        )
        self._packed = None

    @property
    def device(self) -> torch.device:
        return self.tok.device

    @staticmethod
    def _tiles(x3: torch.Tensor) -> torch.Tensor:  # [B,3,224,224] -> [B,256,588], columns (c, y, x): timm PatchEmbed's im2col
        B = x3.shape[0]
        return x3.reshape(B, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(B, 256, 588)

    def embeds(self, pixel_values: torch.Tensor):
        x = pixel_values.to(torch.bfloat16)
        return F.linear(self._tiles(x[:, :3]), self.w0, self.b0), F.linear(self._tiles(x[:, 3:]), self.w1, self.b1)

    def _hidden(self, input_ids, e0, e1):
        x = torch.cat([e0, e1], dim=2).float() @ self.proj  # [B,256,d] fp32 from here on
        te = self.tok[input_ids]
        h = torch.cat([te[:, :1], x, te[:, 1:]], dim=1)
        S = h.shape[1]
        denom = torch.arange(1, S + 1, device=h.device, dtype=h.dtype)[None, :, None]
        c = torch.cumsum(h, dim=1) / denom
        return torch.tanh(c @ self.mix) + 0.25 * h

    def _logits(self, input_ids, e0, e1):
        return self._hidden(input_ids, e0, e1) @ self.head

    def forward(self, input_ids, attention_mask=None, pixel_values=None, labels=None, **_):
        logits = self._logits(input_ids, *self.embeds(pixel_values))
        loss = hf_causal_ce(logits, labels) if labels is not None else None
        return types.SimpleNamespace(loss=loss, logits=logits)

    # ---- the rows / patch-embed interface of OpenVLAShaped ----
    def patch_embed_params(self):
        if not self.w0.is_cuda:
            return None
        if self._packed is None or self._packed[0].device != self.w0.device:
            from . import ops

            self._packed = (ops.pack_embed_weights(self.w0.t().contiguous()), ops.pack_embed_weights(self.w1.t().contiguous()))
        return (self.w0, self.b0, self._packed[0], self.w1, self.b1, self._packed[1])

    def label_row_index(self, labels):
        B, L = labels.shape
        S = N_IMG_TOKENS + L
        bk = (labels[:, 1:] != IGNORE_INDEX).nonzero(as_tuple=False)
        return bk[:, 0] * S + N_IMG_TOKENS + bk[:, 1]

    def forward_rows(self, input_ids, pixel_values, labels, row_index=None, patch_embeds=None, pack=None):
        e0, e1 = (patch_embeds[0], patch_embeds[1]) if patch_embeds is not None else self.embeds(pixel_values)
        logits = self._logits(input_ids, e0, e1)
        if row_index is None:
            row_index = self.label_row_index(labels)
        return logits.reshape(-1, logits.shape[-1]).index_select(0, row_index)


class _BiaslessHead(nn.Module):
    """`lm_head`: F.linear with a frozen [vocab, d] weight (built from the given tensor: nn.Linear's constructor would draw from the global RNG)."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)

    def forward(self, x):
        return F.linear(x, self.weight)


class SurrogateHeadVLA(SurrogateEmbedVLA):
    """SurrogateEmbedVLA with a bf16 LM head over bf16 hidden states — `lm_head.weight` [vocab, d] as LlamaForCausalLM holds it, logits upcast to
    fp32 ([3p transformers 4.40.1 modeling_llama.py]: `logits = logits.float()`) — and the hidden-rows interface of OpenVLAShaped
    (`hidden_rows`, `lm_head`): the smallest model on which the slice-only head K3s (vaa_head_slice_fwd_bwd) runs inside an attack loop, so that
    K3s, K2' and the fused update sit together on a REFERENCE-LOOP trajectory (tools/gen_golden.py:gen_trajectory_upa_k3s drives the
    reference's own UPA loop over this module on the CPU: bf16 matmul head, torch autograd through it).  d = 192: three of K3s's 64-wide
    k-chunks, so its chunk rotation is exercised."""

    def __init__(self, d: int = 192, D0: int = 64, D1: int = 128, vocab: int = MODEL_VOCAB, seed: int = 0):
        super().__init__(d=d, D0=D0, D1=D1, vocab=vocab, seed=seed)
        g = torch.Generator().manual_seed(seed + 7919)
        self.lm_head = _BiaslessHead((torch.randn(vocab, d, generator=g) * (16.0 / math.sqrt(d))).to(torch.bfloat16))
        self.head = None  # the fp32 head of the parent is not part of this model

    def _logits(self, input_ids, e0, e1):
        return self.lm_head(self._hidden(input_ids, e0, e1).to(torch.bfloat16)).float()

    def hidden_rows(self, input_ids, pixel_values, row_index, patch_embeds=None, pack=None):
        e0, e1 = (patch_embeds[0], patch_embeds[1]) if patch_embeds is not None else self.embeds(pixel_values)
        h = self._hidden(input_ids, e0, e1)
        return h.reshape(-1, h.shape[-1]).index_select(0, row_index).to(torch.bfloat16)
