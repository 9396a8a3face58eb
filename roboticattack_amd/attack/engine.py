"""Shared machinery of the UADA / UPA / TMA attack loops (the hot inner step and the validation/checkpoint side).

One inner step = K1 (paste/warp, HIP) -> model forward/backward (PyTorch-ROCm) -> K3 (loss fwd+bwd, HIP) ->
K2 (patch-grad gather, HIP, inside autograd backward) -> [RCCL all-reduce of the 30 KB patch gradient] -> K4 (update, HIP).
Nothing in the step synchronises with the host: the loss scalars of all innerLoop steps of an outer iteration are kept
in a device buffer and read back once (the reference calls `.item()` 4-5 times per inner step, UADA.py:149-154).
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

from .. import _lib, ops
from ..action_tokenizer import ActionTokenizer
from ..constants import MEAN0, MEAN1, STD0, STD1
from ..transform import RandomPatchTransform

try:  # optional, exactly like a reference run with --wandb_project false
    import wandb  # type: ignore
except Exception:  # pragma: no cover
    wandb = None


class NonFiniteAttackState(RuntimeError):
    """The attack state (loss scalars, optimiser moments, patch) stopped being finite, or a kernel reported a device-side failure: raised
    before anything is written to disk. The reference would silently keep optimising and save the patch (UADA.py:257-275)."""


def wandb_enabled(args) -> bool:
    return wandb is not None and args is not None and getattr(args, "wandb_project", "false") != "false"


class ValReadback:
    """What a validation pass reads back from the device — the loss scalars of every batch and, where the metrics need them, the prediction
    and label maps — kept ON the device while the batches are enqueued and fetched ONCE at the end (one [batches, 8] copy + one copy of
    the concatenated maps). The reference calls `.item()` / `.cpu()` several times per batch (UADA.py:229-246); round 5 still read
    `scalars.cpu()` once per batch, i.e. the host waited for every forward before it enqueued the next one.
    VAA_VAL_SYNC_EVERY_BATCH=1 restores the per-batch read-back (A/B measurements). The accumulation below runs in the order of the
    batches on the host, in the reference's precision: the same numbers as the per-batch form."""

    def __init__(self, batches: int, device):
        self.scal = torch.zeros((max(batches, 1), 8), dtype=torch.float32, device=device)
        self.k = 0
        self.maps = []  # per batch: (pred i32 [B,L-1] device, labels[:,1:] i64 device) or None
        self.sync_every = os.environ.get("VAA_VAL_SYNC_EVERY_BATCH", "0") == "1"

    def add(self, scalars, pred=None, labels=None):
        self.scal[self.k].copy_(scalars)
        self.maps.append((pred, labels[:, 1:]) if pred is not None else None)
        self.k += 1
        if self.sync_every:
            self.scal[self.k - 1].cpu()

    def read(self):
        """-> (scalars float64 [batches, 8] on the host, [(pred, gt) numpy int arrays per batch | None])."""
        host = self.scal[: self.k].cpu().numpy().astype(np.float64)
        have = [m for m in self.maps if m is not None]
        out = [None] * len(self.maps)
        if have:
            flat = torch.cat([m[0].reshape(-1).to(torch.int64) for m in have] + [m[1].reshape(-1) for m in have]).cpu().numpy()
            sizes = [int(m[0].numel()) for m in have]
            off, off2, j = 0, sum(sizes), 0
            for i, m in enumerate(self.maps):
                if m is None:
                    continue
                n = sizes[j]
                out[i] = (flat[off : off + n].reshape(tuple(m[0].shape)), flat[off2 : off2 + n].reshape(tuple(m[1].shape)))
                off, off2, j = off + n, off2 + n, j + 1
        return host, out


class AttackBase:
    """State every OpenVLAAttacker variant shares (UADA.py:34-74, UPA.py:31-71, TMA.py:29-64)."""

    val_batches = 1000  # UADA.py:202; subclasses override (UPA/TMA: 100)

    def __init__(self, vla, processor=None, save_dir: str = "", optimizer: str = "pgd", resize_patch: bool = False):
        self.vla = vla.eval()
        self.processor = processor
        tok = getattr(processor, "tokenizer", None)
        self.action_tokenizer = ActionTokenizer(tok)
        self.base_tokenizer = tok
        self.predict_stop_token = True
        self.pad_token_id = 32000
        self.model_max_length = 2048
        self.loss_buffer = []
        self.save_dir = save_dir
        self.device = torch.device(getattr(vla, "device", "cuda"))
        self.randomPatchTransform = RandomPatchTransform(self.device, resize_patch)
        if os.environ.get("VAA_FUSED_EMBED_GRAD", "1") != "0" and hasattr(vla, "forward_rows") and hasattr(vla, "patch_embed_params"):
            # a model that exposes its patch-embed parameters is handed patch-embed OUTPUTS in training steps and K2' never
            # builds the dense pixel gradient (the embed backward runs on the ~20 flagged tiles under the patch only; per-image patches of
            # resize_patch=True included). A black-box model (no patch_embed_params) and VAA_FUSED_EMBED_GRAD=0 keep the pixel_values
            # boundary of the reference.
            self.randomPatchTransform.embed_with = vla
        self.mean = [torch.tensor(MEAN0), torch.tensor(MEAN1)]
        self.std = [torch.tensor(STD0), torch.tensor(STD1)]
        self.optimizer = optimizer
        self.use_rows = hasattr(vla, "forward_rows")  # LM head on labelled rows only (no [B,S,32064] logits)
        self.n_img_tokens = vla.vision_backbone.featurizer.patch_embed.num_patches

    # ---- the model + loss leg of a step ----
    def model_loss(self, input_ids, attention_mask, pix, labels, mode, w=5.0, alpha=0.8, beta=0.2, scale=1.0, need_grad=True, full_ce=True, read_scalars=True):
        """Returns (total [autograd scalar or None], scalars f32[8] device, pred i32 [B,L-1] device).

        `full_ce=False` (slice modes — UADA_DDP, UPA — on a model that exposes its hidden rows): the caller will not read this step's
        full-vocabulary CE (scalars[1]) nor the full argmax (`pred`): the reference reads them on the LAST inner step of an outer iteration
        only (UADA_ddp.py:214-221) and never in UPA's reverse-direction mode (UPA.py:145-186). The step then runs K3s alone — 2.1 MB of head
        weights instead of the 263 MB stream; scalars[1] = 0, pred = -1. The gradient path is the same on every step either way.
        `read_scalars=False` (same models and modes, training steps): the caller reads NONE of this step's loss scalars — the loops read the
        scalars of the last inner step of an outer iteration only (UADA_ddp.py:214-221, UPA.py:171-186) — so not even the slice statistics are
        folded: scalars = 0 (implies full_ce=False).

        `pred` is the argmax over the WHOLE vocabulary at every labelled position (-1 elsewhere), i.e. the reference's
        `action_logits.argmax(dim=2)` (UADA.py:165-167, TMA.py:148-149) that its relative-distance / L1 / ASR metrics and the
        best-patch selection read; the action-slice argmax of UADA.py:395 only feeds UAD, which K3 reports in scalars[7]."""
        if self.use_rows:
            pack = self._rows_cache(labels, attention_mask)
            pe = pix if isinstance(pix, ops.PatchEmbeds) else None  # patched batch handed over as patch-embed outputs (pixel gradient never built)
            if need_grad and hasattr(self.vla, "hidden_rows"):
                # LM head + loss on the labelled rows: the head's backward contracts over the 256 action
                # columns when the loss lives there (UADA_DDP, UPA)
                h = self.vla.hidden_rows(input_ids, None if pe is not None else pix, self._row_index, patch_embeds=pe, pack=pack)
                W = self.vla.lm_head.weight
                # ... and with the LM head FUSED into K3's statistics there (K3h: no [R,V] logits at all); UADA's 1/CE and TMA's CE gradients need
                # every logit and keep the GEMM head
                if mode in ops.SLICE_MODES and self._slice_head(int(h.shape[0]), h, W):
                    # K3s: slice-only head + statistics + gradient + head backward in ONE launch, on every step; K3h's 263 MB stream only
                    # behind it on the steps whose CE / full argmax is read
                    total, scalars, _, pred_full = ops.HeadSliceLoss.apply(h, W, self._row_map, mode, w, alpha, beta, scale, bool(full_ce), bool(read_scalars))
                    return total, scalars, pred_full
                head = ops.HeadLossRowsFused if (mode in ops.SLICE_MODES and self._fused_head(int(h.shape[0]), h, W)) else ops.HeadLossRows
                total, scalars, _, pred_full = head.apply(h, W, self._row_map, mode, w, alpha, beta, scale)
                return total, scalars, pred_full
            if not need_grad and hasattr(self.vla, "hidden_rows") and self._row_count > 0 and self._fused_head(self._row_count, None, self.vla.lm_head.weight):
                # evaluation only (validation passes, every mode): head + statistics + fold without logits in memory when K3h covers the shape
                h = self.vla.hidden_rows(input_ids, None if pe is not None else pix, self._row_index, patch_embeds=pe, pack=pack)
                if not full_ce and mode in ops.SLICE_MODES and self._slice_head(self._row_count, h, self.vla.lm_head.weight):
                    o = ops.head_slice_fwd_bwd(h.detach().contiguous(), self.vla.lm_head.weight, self._row_map, mode, w, alpha, beta, scale, want_dh=False)
                    return None, o["scalars"], o["pred_full"]
                if h.dtype == torch.bfloat16:
                    scalars, _, pred_full, _ = ops.head_loss_rows_fwd_bwd(h.detach().contiguous(), self.vla.lm_head.weight, self._row_map, mode, w, alpha, beta,
                                                                          scale, want_grad=False)
                    return None, scalars, pred_full
                logits = self.vla.lm_head(h)
                scalars, _, pred_full, _ = ops.loss_rows_fwd_bwd(logits.detach().contiguous(), self._row_map, mode, w, alpha, beta, scale, want_grad=False)
                return None, scalars, pred_full
            logits = self.vla.forward_rows(input_ids, None if pe is not None else pix, labels, self._row_index, patch_embeds=pe, pack=pack)
            if need_grad:
                total, scalars, _, pred_full = ops.DiscrepancyLossRows.apply(logits.contiguous(), self._row_map, mode, w, alpha, beta, scale)
                return total, scalars, pred_full
            scalars, _, pred_full, _ = ops.loss_rows_fwd_bwd(logits.detach().contiguous(), self._row_map, mode, w, alpha, beta, scale, want_grad=False)
            return None, scalars, pred_full
        out = self.vla(input_ids=input_ids, attention_mask=attention_mask, pixel_values=pix, labels=None)
        logits = out.logits
        if need_grad:
            total, scalars, _, pred_full = ops.DiscrepancyLoss.apply(logits.contiguous(), labels, mode, w, alpha, beta, scale, ops.LAYOUT_FULL)
            return total, scalars, pred_full
        scalars, _, _, pred_full = ops.loss_fwd_bwd(logits.detach().contiguous(), labels, mode, w, alpha, beta, scale, ops.LAYOUT_FULL,
                                                    want_grad=False, want_pred_full=True)
        return None, scalars, pred_full

    def _rows_cache(self, labels, attention_mask):
        """labels are fixed during an outer iteration: the row index (one host sync) and the device row map of K3 are cached per
        tensor OBJECT; the cache holds a reference, so identity cannot be recycled by the allocator, and in-place edits bump _version.
        Returns the sequence pack (or None)."""
        if getattr(self, "_row_ref", None) is not labels or self._row_ver != labels._version:
            self._row_ref, self._row_ver, self._row_index = labels, labels._version, self.vla.label_row_index(labels)
            self._row_map = ops.LossRowMap(labels)
            self._row_count = int(self._row_index.numel())
        pack = None
        if hasattr(self.vla, "make_pack") and attention_mask is not None:  # drop the padding rows (cached like the row index)
            if getattr(self, "_pack_ref", None) is not attention_mask or self._pack_ver != attention_mask._version:
                self._pack_ref, self._pack_ver, self._pack = attention_mask, attention_mask._version, self.vla.make_pack(attention_mask)
            pack = self._pack
        return pack

    # ---- the data-parallel UADA step with the fused epilogue ----
    def fused_ddp_available(self) -> bool:
        """K1 (tile-major) -> model -> K3 statistics (+ gradient slice) -> K2' tiles + scatter -> ONE epilogue launch (K2's final sum, K3's
        fold, the DDP message) needs a model that exposes its patch-embed weights and its hidden rows, and one patch per batch."""
        t = self.randomPatchTransform
        return (os.environ.get("VAA_FUSED_EPILOGUE", "1") != "0" and self.use_rows and hasattr(self.vla, "hidden_rows")
                and t.embed_with is not None and not t.resize_patch)

    @staticmethod
    def _fused_head(R, h, W) -> bool:
        """LM head fused with K3's statistics (vaa_head_loss_rows_stats) for this shape? Whenever the kernel covers it (bf16, up to 128 labelled
        rows = bs 64 with maskidx=[0]): one pass over the 263 MB head weight, the [R,V] logits are never written. In the step, same box:
        bs = 8 (16 rows) 45 us (5.8 TB/s) + 7 us fold against 64 + 10 us for the hipBLASLt GEMM + K3 statistics; bs = 32: 47 + 8 against
        62 + 10; bs = 64 (128 rows, where a CU's line rate, not HBM, bounds the kernel): 54 + 8 against 55-57 + 11
        (profiles/r04_head_fused.txt). VAA_FUSED_HEAD=0 restores the GEMM path."""
        mode = os.environ.get("VAA_FUSED_HEAD", "auto")
        if mode == "0" or (h is not None and h.dtype != torch.bfloat16) or W.dtype != torch.bfloat16 or not W.is_cuda:
            return False
        return ops.head_loss_rows_applies(R, int(W.shape[1]), int(W.shape[0]))

    @staticmethod
    def _slice_head(R, h, W) -> bool:
        """K3s (vaa_head_slice_fwd_bwd) for this shape? bf16, up to 128 labelled rows, D a multiple of 64 up to 4096. VAA_HEAD_EVERY_STEP=1
        restores the round-5 behaviour (K3h + finish + the 256-column GEMM on every step), VAA_FUSED_HEAD=0 the GEMM head."""
        if os.environ.get("VAA_HEAD_EVERY_STEP", "0") == "1" or os.environ.get("VAA_FUSED_HEAD", "auto") == "0":
            return False
        if (h is not None and h.dtype != torch.bfloat16) or W.dtype != torch.bfloat16 or not W.is_cuda:
            return False
        return ops.head_slice_applies(R, int(W.shape[1]), int(W.shape[0]))

    def fused_ddp_step(self, pixel_values, patch, input_ids, attention_mask, labels, geometry, w, msg, scalars, optimizer=None, full_ce=True):
        """The UADA_ddp inner step (UADA_ddp.py:189-206) up to the gradient exchange, six hand-written launches around the model:
        K1 -> [ViTs, Llama, LM head on the labelled rows] -> K3 statistics + gradient slice -> [head / model backward] -> K2' tile GEMM ->
        scatter -> epilogue. On return `msg` (f32 [3*ph*pw + 4]) holds [patch gradient | CE, w^2*MSE, UAD, total] of THIS rank, ready for
        one all-reduce, and `scalars` (f32[8]) the loss scalars; returns the full-vocabulary predictions [B,L-1] (i32, device) — all three only on
        the steps the caller marks `full_ce` (the steps whose scalars are read; else the tail / `scalars` are left as they were and None is returned).
        Nothing is synchronised; patch.grad is not touched. `optimizer` (single-GPU run, no L1 clip): K4 is applied by the epilogue launch
        itself — five launches per step, the caller then calls neither the all-reduce nor optimizer.step()."""
        pack = self._rows_cache(labels, attention_mask)
        if self._row_count == 0:
            raise ValueError("fused_ddp_step: no labelled position in the batch")
        sink = {}
        pe = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch, mean=self.mean, std=self.std, geometry=geometry, grad_sink=sink)
        h = self.vla.hidden_rows(input_ids, None, self._row_index, patch_embeds=pe, pack=pack)
        W = self.vla.lm_head.weight
        R = int(h.shape[0])
        if self._slice_head(R, h, W):
            # K3s: slice logits -> statistics -> gradient slice -> d total / d hidden in ONE launch (2.1 MB of head weights); the full-vocabulary
            # stream (K3h, 263 MB) runs behind it only on the steps whose CE / full argmax is read (`full_ce`: UADA_ddp.py:214-221)
            hd = h.detach().contiguous()
            o = ops.head_slice_fwd_bwd(hd, W, self._row_map, ops.LOSS_UADA_DDP, w, want_dh=True, want_scalars=False)
            ws = ops.head_loss_rows_stats(hd, W, self._row_map, ops.LOSS_UADA_DDP, w) if full_ce else o["ws"]
            h.backward(o["dh"])
            upd = optimizer.fused_update_args() if optimizer is not None else None
            if not full_ce:
                # nobody reads this step's loss scalars (the loop reads those of the LAST inner step of an outer iteration: UADA_ddp.py:214-221):
                # the epilogue in its pass-through form — K2's final sum (+ K4), no fold; `scalars` / the message tail keep the last read step's values
                ops.step_epilogue(sink["partials"], msg, scalars, update=upd)
                return None
            _, pred_full = ops.step_epilogue(sink["partials"], msg, scalars, rowmap=self._row_map, R=R, V=int(W.shape[0]),
                                             mode=ops.LOSS_UADA_DDP, w=w, loss_ws=ws, update=upd)
            return pred_full
        gsl = torch.empty((R, ops.N_ACTION), dtype=h.dtype, device=h.device)
        if self._fused_head(R, h, W):
            # SURVEY.md 8f-2 as the survey wrote it: LM head + K3 statistics in ONE weight-streaming kernel — the [R,V] logits are never written
            ws = ops.head_loss_rows_stats(h.detach().contiguous(), W, self._row_map, ops.LOSS_UADA_DDP, w, grad=gsl)
        else:
            logits = torch.nn.functional.linear(h.detach(), W)  # [R,V]: the LM head on the labelled rows through hipBLASLt
            ws = ops.loss_rows_stats(logits, self._row_map, ops.LOSS_UADA_DDP, w, grad=gsl)  # K3: statistics + d loss / d action logits
        h.backward(gsl @ W[ops.ACTION_LO : ops.ACTION_LO + ops.N_ACTION])                     # head backward over 256 columns, model backward, K2'
        _, pred_full = ops.step_epilogue(sink["partials"], msg, scalars, rowmap=self._row_map, R=R, V=int(W.shape[0]),
                                         mode=ops.LOSS_UADA_DDP, w=w, loss_ws=ws,
                                         update=optimizer.fused_update_args() if optimizer is not None else None)
        return pred_full

    # ---- single-GPU loops: K2's final sum + K4 as one launch after the backward ----
    def fused_update_sink(self, optimizer):
        """A gradient sink ({}) for `apply_random_patch_batch(grad_sink=...)` when the step can end with ONE launch that adds K2''s partial tiles
        and applies the optimiser to every gradient element as it is produced (`fused_update`), else None: needs one patch per batch and no
        L1 clip (UPA's clip needs the whole gradient's norm first). Works behind both boundaries: K2' (a model that exposes its patch-embed
        weights) and K2 on a black-box model's pixel gradient."""
        t = self.randomPatchTransform
        ok = os.environ.get("VAA_FUSED_EPILOGUE", "1") != "0" and not t.resize_patch and not optimizer.l1_clip
        return {} if ok else None

    def fused_update(self, sink, patch, optimizer, scalars):
        """After `total.backward()` of a step whose transform was given `sink`: vaa_step_epilogue_update in its pass-through form (the loss
        scalars are final already) — patch / m / v get the bits `optimizer.step()` would write; patch.grad stays None."""
        n = patch.numel()
        msg = getattr(self, "_epi_msg", None)
        if msg is None or msg.numel() != n + 4 or msg.device != patch.device:
            msg = self._epi_msg = torch.zeros(n + 4, dtype=torch.float32, device=patch.device)
        ops.step_epilogue(sink["partials"], msg, scalars, update=optimizer.fused_update_args())

    # ---- fail loud, never NaN: once per outer iteration, behind the read-back that synchronises anyway ----
    def assert_finite_state(self, patch, optimizer, host_scalars, where: str, all_ranks: bool = False, device_failure=None):
        """`host_scalars`: the loss scalars of the outer iteration's inner steps, already on the host (their read-back was the sync).
        AdamW's m / v are the sticky witnesses of any non-finite gradient of ANY inner step (a NaN that entered them never leaves, while the
        clamp turns the patch itself into a finite 0), the library's failure word covers kernels that had to give up (vaa_async_error: a
        per-PROCESS word). `all_ranks` (the data-parallel loop): the verdict is all-reduced (MIN) before anyone raises, so that every rank
        leaves the loop together instead of one rank raising while the others wait in the next all-reduce until the RCCL timeout."""
        # `device_failure`: a failure this rank's inner loop already caught (and polled) — the data-parallel loop keeps its collectives going and
        # hands it over here, so that the verdict below is all-reduced instead of one rank raising alone
        try:
            ops.async_error_check()
        except _lib.VaaError as e:  # reported like every other non-finite state (and cleared: the poll is the consumer of the sticky word)
            device_failure = device_failure or str(e)
        ok_dev = torch.isfinite(patch.detach()).all()
        if getattr(optimizer, "m", None) is not None:
            ok_dev = ok_dev & torch.isfinite(optimizer.m).all() & torch.isfinite(optimizer.v).all()
        stats = getattr(optimizer, "last_stats", None)
        if stats is not None:
            ok_dev = ok_dev & torch.isfinite(stats).all()
        ok_host = bool(np.isfinite(np.asarray(host_scalars, dtype=np.float64)).all())
        ok_local = ok_host and bool(ok_dev) and device_failure is None
        ok = ok_local
        if all_ranks:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                flag = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float32, device=patch.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item() > 0.5)
        if not ok:
            raise NonFiniteAttackState(f"{where}: non-finite attack state (loss scalars finite: {ok_host}; patch / moments / gradient statistics "
                                       f"finite: {bool(ok_dev)}; device-side failure: {device_failure or 'none'}"
                                       + ("" if ok_local else "; detected on THIS rank") + (" — raised on every rank" if all_ranks else "")
                                       + ") — nothing was saved for this iteration")

    # ---- metrics (host, once per outer iteration) ----
    def decode_pred_gt(self, pred: torch.Tensor, labels: torch.Tensor):
        """Continuous predicted / ground-truth actions of the action rows, (b,k) order (UADA.py:165-175)."""
        p = pred.detach().cpu().numpy()
        gt = labels[:, 1:].detach().cpu().numpy()
        m = gt > self.action_tokenizer.action_token_begin_idx
        return (torch.tensor(self.action_tokenizer.decode_token_ids_to_actions(p[m])),
                torch.tensor(self.action_tokenizer.decode_token_ids_to_actions(gt[m])))

    def decode_pred_gt_np(self, p: np.ndarray, gt: np.ndarray):
        """decode_pred_gt on maps that are already on the host (ValReadback.read): p [B,L-1] predictions, gt = labels[:, 1:]."""
        m = gt > self.action_tokenizer.action_token_begin_idx
        return (torch.tensor(self.action_tokenizer.decode_token_ids_to_actions(p[m])),
                torch.tensor(self.action_tokenizer.decode_token_ids_to_actions(gt[m])))

    def calculate_relative_distance(self, pred, gt, maskidx, relative_distance):
        """UADA.py:354-369."""
        pred = pred.clone().view(pred.shape[0] // len(maskidx), len(maskidx))
        gt = gt.clone().view(gt.shape[0] // len(maskidx), len(maskidx))
        for i1 in range(pred.shape[0]):
            for i2 in range(pred.shape[1]):
                anchor = gt[i1, i2]
                max_boundary = max(1 - anchor, anchor - (-1))
                relative_distance[f"{str(maskidx[i2])}"].append((abs(pred[i1, i2] - anchor) / max_boundary).item())
        return relative_distance

    def filter_train(self, data):
        """UADA.py:308-339 — optional gripper filtering; keeps the reference's thresholds."""
        import random

        pixel_values = data["pixel_values"]
        labels = data["labels"].to(self.device)
        attention_mask = data["attention_mask"].to(self.device)
        input_ids = data["input_ids"].to(self.device)
        masked = labels[labels > self.action_tokenizer.action_token_begin_idx]
        masked = masked.view(masked.shape[0] // 7, 7)
        one_index = [i for i in range(masked.shape[0]) if masked[i, 6] == 31744]
        chosen = None
        if 1 < len(one_index) < 8:
            chosen = one_index
        elif len(one_index) > 8:
            chosen = random.sample(one_index, k=8)
        if chosen is not None:
            labels, attention_mask, input_ids = labels[chosen, :], attention_mask[chosen, :], input_ids[chosen, :]
            pixel_values = [pixel_values[i] for i in chosen]
        return labels, attention_mask, input_ids, pixel_values

    # ---- outputs (a-12) ----
    def save_patch(self, patch: torch.Tensor, sub: str) -> str:
        """`torch.save(patch.detach().cpu(), <save_dir>/<sub>/patch.pt)` — plain fp32 [3,ph,pw] CPU tensor (UADA.py:257-275)."""
        d = os.path.join(self.save_dir, sub)
        host = patch.detach().float().cpu().contiguous().clone()
        if not bool(torch.isfinite(host).all()):  # belt and braces behind assert_finite_state
            raise NonFiniteAttackState(f"save_patch({sub!r}): the patch holds non-finite values")
        os.makedirs(d, exist_ok=True)
        torch.save(host, os.path.join(d, "patch.pt"))
        return d

    def save_val_images(self, modified_images: torch.Tensor, d: str):
        """De-normalised first-3-channel frames as PNG (UADA.py:260-268): ToPILImage == mul(255).byte()."""
        from PIL import Image

        path = os.path.join(d, "val_related_data")
        os.makedirs(path, exist_ok=True)
        imgs = self.randomPatchTransform.denormalize(modified_images[:, 0:3].detach().float().cpu(), self.mean[0], self.std[0])
        pil = []
        for o in range(imgs.shape[0]):
            arr = imgs[o].mul(255).byte().permute(1, 2, 0).numpy()
            im = Image.fromarray(arr)
            im.save(os.path.join(path, f"{o}.png"))
            pil.append(im)
        return path, pil

    def plot_loss(self):
        """UADA.py:76-91 (seaborn theme dropped; the curve is identical)."""
        try:
            import matplotlib

            matplotlib.use("Agg")
            import matplotlib.pyplot as plt

            plt.plot(list(range(len(self.loss_buffer))), self.loss_buffer, label="Target Loss")
            plt.title("Loss Plot")
            plt.xlabel("Iters")
            plt.ylabel("Loss")
            plt.legend(loc="best")
            plt.savefig("%s/loss_curve.png" % (self.save_dir))
            plt.clf()
        except Exception:  # plotting is best effort
            pass
        torch.save(self.loss_buffer, "%s/loss" % (self.save_dir))

    def dump_lists(self, names):
        for n in names:
            with open(os.path.join(self.save_dir, f"{n}.pkl"), "wb") as f:
                pickle.dump(getattr(self, n), f)


def to_dev(batch, device):
    """labels/attention_mask/input_ids .to(device) (UADA.py:125-127); `.to` on a same-device tensor aliases, and
    mask_labels mutates in place, so labels are cloned to keep the loader's copy intact."""
    return (batch["pixel_values"], batch["labels"].to(device).clone(), batch["attention_mask"].to(device),
            batch["input_ids"].to(device))


def next_or_restart(iterator, loader):
    try:
        return next(iterator), iterator
    except StopIteration:
        iterator = iter(loader)
        return next(iterator), iterator
