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
