"""Patch optimiser + LR schedule, host side. The arithmetic runs in ONE fused HIP launch (K4, include/vaa.h:vaa_patch_update).

Mirrors what the attack loops use from `transformers` [3p, ==4.40.1 in the reference's pyproject.toml:50]:
  * `transformers.AdamW([patch], lr)` (UADA.py:108, UADA_ddp.py:167, UPA.py:108, TMA.py:102): betas (0.9, 0.999),
    eps 1e-6 added to sqrt(v) before the bias-corrected step size is applied, weight_decay 0, correct_bias True.
    That class no longer exists in transformers 5.x; the restated algorithm is "parity unpinned" (SURVEY.md §8c).
  * `transformers.get_cosine_schedule_with_warmup(opt, warmup, total, num_cycles=0.5)` (UADA.py:109-115), stepped once
    per OUTER iteration (UADA.py:162-164) — so every inner step of outer iteration 0 runs with lr = 0.
The object exposes the small surface the loops touch: `.step()`, `.zero_grad()`, `.param_groups[0]["lr"]`.
"""
from __future__ import annotations

import math

import torch

from . import ops


def cosine_with_warmup_lambda(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class PatchOptimizer:
    """Owns m, v and the step count of ONE patch tensor; `step()` = K4 (scale, clip, AdamW|PGD, clamp to [0,1])."""

    def __init__(self, patch: torch.Tensor, lr: float, mode: str = "adamW", betas=(0.9, 0.999), eps: float = 1e-6,
                 l1_clip: float = 0.0, clamp: bool = True):
        if mode not in ("adamW", "pgd"):
            raise ValueError(f"unknown optimizer {mode!r}")
        self.patch = patch
        self.mode = ops.OPT_ADAMW_HF if mode == "adamW" else ops.OPT_PGD_SIGN
        self.param_groups = [dict(params=[patch], lr=lr, initial_lr=lr, betas=betas, eps=eps)]
        self.l1_clip = l1_clip
        self.t = 0
        self.m = torch.zeros_like(patch, requires_grad=False) if self.mode == ops.OPT_ADAMW_HF else None
        self.v = torch.zeros_like(patch, requires_grad=False) if self.mode == ops.OPT_ADAMW_HF else None
        self._last_stats = None  # device f32[2]: [sum|g|, mean g] of the most recent step
        self._stat_part = None   # fused steps (vaa_step_epilogue_update): per-block {sum|g|, sum g}, folded on demand

    @property
    def last_stats(self):
        """device f32[2] = [sum|g|, mean g] of the most recent step (what K4 logs; after a fused step folded here from its block sums)."""
        if self._last_stats is None and self._stat_part is not None:
            t = self._stat_part.sum(dim=0)
            self._last_stats = torch.stack([t[0], t[1] / self.patch.numel()]).to(torch.float32)
        return self._last_stats

    @last_stats.setter
    def last_stats(self, v):
        self._last_stats = v

    def fused_update_args(self):
        """Arguments of ONE step for ops.step_epilogue(update=...) — the single-GPU step applies K4 inside the epilogue launch. Advances the
        step count like step(); only without L1 clip (the clip needs the whole gradient's norm first)."""
        if self.l1_clip:
            raise ValueError("the fused update has no L1 clip")
        self.t += 1
        grp = self.param_groups[0]
        if self._stat_part is None:
            self._stat_part = torch.zeros(((self.patch.numel() + 63) // 64, 2), dtype=torch.float64, device=self.patch.device)
        self._last_stats = None
        return dict(patch=self.patch.data, m=self.m, v=self.v, mode=self.mode, lr=grp["lr"], step=self.t, beta1=grp["betas"][0],
                    beta2=grp["betas"][1], eps=grp["eps"], stat_part=self._stat_part)

    def step(self, grad: torch.Tensor | None = None, grad_scale: float = 1.0):
        g = grad if grad is not None else self.patch.grad
        if g is None:
            return None
        self.t += 1
        grp = self.param_groups[0]
        self._last_stats = ops.patch_update(self.patch.data, g.contiguous(), self.m, self.v, self.mode, grp["lr"], self.t,
                                            grp["betas"][0], grp["betas"][1], grp["eps"], self.l1_clip, grad_scale)
        return self._last_stats

    def zero_grad(self, set_to_none: bool = True):
        if set_to_none:
            self.patch.grad = None
        elif self.patch.grad is not None:
            self.patch.grad.zero_()


class CosineWarmupSchedule:
    """LambdaLR semantics: lr = initial_lr * lambda(epoch); constructed at epoch 0, `.step()` advances by one."""

    def __init__(self, optimizer: PatchOptimizer, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
        self.opt, self.w, self.T, self.c = optimizer, num_warmup_steps, num_training_steps, num_cycles
        self.last_epoch = 0
        self._apply()

    def _apply(self):
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * cosine_with_warmup_lambda(self.last_epoch, self.w, self.T, self.c)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]
