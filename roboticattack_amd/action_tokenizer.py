"""Action (de)tokenisation: host-side mirror of prismatic/vla/action_tokenizer.py:13-72.

256 uniform bins over [-1, 1]; bin index d in [1,256] maps to token id `vocab_size - d`, i.e. the last 256 ids of
the 32000-entry Llama vocabulary (31744..31999). Only the numeric behaviour is mirrored; the text round trip the
reference makes through the Llama tokenizer (TMA.py:93) is the identity on these ids.
"""
from __future__ import annotations

import numpy as np

from .constants import N_BINS, TOKENIZER_VOCAB


class ActionTokenizer:
    def __init__(self, tokenizer=None, bins: int = N_BINS, min_action: int = -1, max_action: int = 1):
        self.tokenizer = tokenizer
        self.vocab = int(getattr(tokenizer, "vocab_size", TOKENIZER_VOCAB))
        self.n_bins, self.min_action, self.max_action = bins, min_action, max_action
        self.bins = np.linspace(min_action, max_action, self.n_bins)  # action_tokenizer.py:31
        self.bin_centers = (self.bins[:-1] + self.bins[1:]) / 2.0  # :32
        self.action_token_begin_idx = int(self.vocab - (self.n_bins + 1))  # :36  (31743)

    def action_to_token_ids(self, action) -> np.ndarray:
        """Numeric content of ActionTokenizer.__call__ (:38-47): clip, digitize, vocab_size - bin."""
        a = np.clip(np.asarray(action, dtype=np.float64), a_min=float(self.min_action), a_max=float(self.max_action))
        return (self.vocab - np.digitize(a, self.bins)).astype(np.int64)

    def __call__(self, action):
        ids = self.action_to_token_ids(action)
        if self.tokenizer is not None and hasattr(self.tokenizer, "decode"):
            return self.tokenizer.decode(list(ids)) if ids.ndim == 1 else self.tokenizer.batch_decode(ids.tolist())
        return ids

    def decode_token_ids_to_actions(self, action_token_ids: np.ndarray) -> np.ndarray:
        """:49-68 — bin centres, with index 255 folded onto the last interval."""
        d = self.vocab - np.asarray(action_token_ids)
        d = np.clip(d - 1, a_min=0, a_max=self.bin_centers.shape[0] - 1)
        return self.bin_centers[d]

    @property
    def vocab_size(self) -> int:
        return self.n_bins
