"""Host-side label handling of the attack loops (int64 index work, once per outer iteration).

mask_labels       : UADA.py:371-379, UADA_ddp.py:89-97, UPA.py:344-356
tma_target_tokens : TMA.py:93-99
tma_target_labels : TMA.py:124-129
"""
from __future__ import annotations

import numpy as np
import torch

from .action_tokenizer import ActionTokenizer
from .constants import ACTION_TOKEN_BEGIN_IDX, EOS_ID, IGNORE_INDEX


def mask_labels(labels: torch.Tensor, maskidx) -> torch.Tensor:
    """Keep only the `maskidx` DoFs among the 7 action tokens of every row (others -> -100); EOS stays. In place,
    like the reference. Raises if a row does not hold exactly 7 action tokens (the reference's view(-1, 7) would)."""
    is_act = labels > ACTION_TOKEN_BEGIN_IDX
    if not bool((is_act.sum(dim=1) == 7).all()):
        raise RuntimeError("mask_labels: every row must contain exactly 7 action tokens")
    keep = torch.zeros(7, dtype=torch.bool, device=labels.device)
    if len(maskidx):
        keep[torch.as_tensor(list(maskidx), dtype=torch.long, device=labels.device)] = True
    ordinal = (torch.cumsum(is_act.to(torch.int64), dim=1) - 1).clamp_(0, 6)
    labels[is_act & ~keep[ordinal]] = IGNORE_INDEX
    return labels


def tma_target_tokens(target_action, maskidx, action_tokenizer: ActionTokenizer | None = None) -> torch.Tensor:
    """The 8-entry target vector: 7 target action tokens (non-maskidx entries -100) followed by EOS (-100 unless 7 in
    maskidx, exactly as TMA.py:97-99 applies `idx not in maskidx` to all 8 positions)."""
    at = action_tokenizer or ActionTokenizer()
    ids = list(at.action_to_token_ids(np.asarray(target_action, dtype=np.float64))) + [EOS_ID]
    t = torch.tensor(ids, dtype=torch.int64)
    for idx in range(len(t)):
        if idx not in maskidx:
            t[idx] = IGNORE_INDEX
    return t


def tma_target_labels(labels: torch.Tensor, target_tokens: torch.Tensor) -> torch.Tensor:
    """Overwrite the 8 non-ignored label positions of every row with the target vector (new tensor)."""
    new = labels.clone()
    sel = new != IGNORE_INDEX
    if not bool((sel.sum(dim=1) == target_tokens.numel()).all()):
        raise RuntimeError("tma_target_labels: every row must contain exactly 8 labelled positions")
    new[sel] = target_tokens.to(new.device).repeat(new.shape[0])
    return new
