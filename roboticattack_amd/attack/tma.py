"""TMA — targeted manipulation attack. Mirrors VLAAttacker/white_patch/TMA.py:28-483.

Loss = HF mean CE against a fixed target-token vector on the `maskidx` DoFs, divided by accumulate_steps (TMA.py:148);
AdamW (+cosine) or PGD sign step (TMA.py:164-175); without geometry the patch is pasted by `paste_patch_fix`
(mask rule canvas != -100, TMA.py:133-135), with geometry by `apply_random_patch_batch` (the `colorjitter=` kwarg the
reference passes is ignored, Appendix A-D3). Validation: 100 batches, ASR / L1 metrics, best patch by L1 (TMA.py:202-383).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import ops
from ..labels import tma_target_labels, tma_target_tokens
from ..optim import CosineWarmupSchedule, PatchOptimizer
from .engine import AttackBase, ValReadback, next_or_restart, to_dev, wandb, wandb_enabled


class OpenVLAAttacker(AttackBase):
    val_batches = 100  # TMA.py:211

    def __init__(self, vla, processor=None, save_dir="", optimizer="pgd", resize_patch=False):
        super().__init__(vla, processor, save_dir, optimizer, resize_patch)
        self.adv_action_L1_loss = []
        self.min_val_avg_CE_loss = 1000000
        self.min_val_avg_L1_loss = 1000000

    def _images(self, pixel_values, patch, geometry, colorjitter, grad_sink=None):
        if not geometry and not colorjitter:
            return self.randomPatchTransform.paste_patch_fix(pixel_values, patch, mean=self.mean, std=self.std)
        kw = {"grad_sink": grad_sink} if grad_sink is not None else {}
        return self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch, mean=self.mean, std=self.std,
                                                                   geometry=geometry, colorjitter=colorjitter, **kw)

    def calculate_relative_distance_target(self, pred, gt):
        """TMA.py:470-483."""
        if pred.shape[0] == 0:
            return torch.tensor(0.0)
        max_b = torch.maximum(1 - gt, gt - (-1))
        return (torch.abs(pred - gt) / max_b).sum() / pred.shape[0]

    def calculate_01_ASR(self, pred, gt):
        """TMA.py:398-420 (gripper ASR bookkeeping)."""
        s02 = n0 = s12 = n1 = so2 = no = 0
        for idx in range(gt.shape[0]):
            if gt[idx] == 31872:
                n0 += 1
                s02 += int(pred[idx] != 31872)
            elif gt[idx] == 31744:
                n1 += 1
                s12 += int(pred[idx] != 31744)
            else:
                no += 1
                so2 += int(pred[idx] == 31872)
        return s02, n0, s12, n1, so2, no

    def inner_step(self, patch, optimizer, pixel_values, input_ids, attention_mask, newlabels, geometry, colorjitter, scalars_out, k,
                   accumulate_steps=1, do_step=True):
        """One iteration of the hot inner loop (TMA.py:131-175): K1 -> model -> GEMM head + K3 (CE gradient over every logit of the labelled rows) ->
        backward -> K2 / K2' -> K4. Without gradient accumulation the step ends with ONE launch: K2's final sum + the optimiser (AdamW or PGD
        sign step) + clamp. Returns (full-vocabulary predictions, whether the update ran inside that launch)."""
        sink = self.fused_update_sink(optimizer) if (accumulate_steps == 1 and geometry) else None
        pix = self._images(pixel_values, patch, geometry, colorjitter, grad_sink=sink)
        total, scalars, pred = self.model_loss(input_ids, attention_mask, pix, newlabels, ops.LOSS_CE, scale=1.0 / accumulate_steps)
        total.backward()
        fused = sink is not None and "partials" in sink
        if fused:
            self.fused_update(sink, patch, optimizer, scalars)
            optimizer.zero_grad()
        elif do_step:
            scalars_out[k, 8:10] = optimizer.step()
            optimizer.zero_grad()
        scalars_out[k, :8] = scalars
        return pred, fused

    def patchattack_unconstrained(self, train_dataloader, val_dataloader, num_iter=5000, target_action=np.zeros(7),
                                  patch_size=[3, 50, 50], alpha=1 / 255, accumulate_steps=1, maskidx=[], warmup=20,
                                  filterGripTrainTo1=False, geometry=False, colorjitter=False, innerLoop=1, args=None):
        self.val_CE_loss, self.val_L1_loss, self.val_ASR, self.val_inner_relatived_distance = [], [], [], []
        self.train_CE_loss, self.train_inner_avg_loss, self.train_inner_relatived_distance = [], [], []
        dev = self.device
        patch = torch.rand(patch_size).to(dev)
        patch.requires_grad_(True)
        self.patch = patch
        target = tma_target_tokens(target_action, maskidx, self.action_tokenizer).to(dev)  # TMA.py:93-99
        print(f"target_action: {target}")
        optimizer = PatchOptimizer(patch, alpha, "adamW" if self.optimizer == "adamW" else "pgd")
        scheduler = CosineWarmupSchedule(optimizer, warmup, int(num_iter / accumulate_steps), 0.5) if self.optimizer == "adamW" else None
        train_iterator, val_iterator = iter(train_dataloader), iter(val_dataloader)
        scal = torch.zeros((max(innerLoop, 1), 10), dtype=torch.float32, device=dev)

        for i in range(num_iter):
            data = next(train_iterator)
            if len(maskidx) == 1 and maskidx[0] == 6 and filterGripTrainTo1:
                labels, attention_mask, input_ids, pixel_values = self.filter_train(data)
            else:
                pixel_values, labels, attention_mask, input_ids = to_dev(data, dev)
            newlabels = tma_target_labels(labels, target)  # TMA.py:124-129
            do_step = (i + 1) % accumulate_steps == 0 or (i + 1) == len(train_dataloader)
            rel = []
            fused_row = None
            for inner_loop in range(innerLoop):
                pred, fused = self.inner_step(patch, optimizer, pixel_values, input_ids, attention_mask, newlabels, geometry, colorjitter, scal, inner_loop,
                                              accumulate_steps=accumulate_steps, do_step=do_step)
                if fused:
                    fused_row = inner_loop
                rel.append(pred)
            if scheduler is not None and do_step:
                scheduler.step()
            if fused_row is not None:
                scal[fused_row, 8:10] = optimizer.last_stats
            host = scal[:innerLoop].cpu().numpy()
            self.assert_finite_state(patch, optimizer, host[:, :8], f"TMA outer iteration {i}")
            inner_avg_loss = float(host[:, 0].mean())
            inner_rel = 0.0
            for p in rel:  # TMA.py:153-162
                cp, cg = self.decode_pred_gt(p, newlabels)
                inner_rel += float(self.calculate_relative_distance_target(cp, cg))
            inner_rel /= innerLoop
            loss = float(host[-1, 0])
            self.loss_buffer.append(loss)
            print(f"target_loss: {loss}")
            self.last_train_log = {"TRAIN_attack_loss(CE)": loss, "TRAIN_patch_gradient": float(host[-1, 9]),
                                   "TRAIN_LR": optimizer.param_groups[0]["lr"], "TRAIN_inner_avg_loss": inner_avg_loss,
                                   "TRAIN_inner_relatived_distance": inner_rel}
            if wandb_enabled(args):
                wandb.log(self.last_train_log, step=i)
            self.train_CE_loss.append(loss)
            self.train_inner_avg_loss.append(inner_avg_loss)
            self.train_inner_relatived_distance.append(inner_rel)
            if i % 100 == 0:
                self.plot_loss()
                val_iterator = self.validate(i, patch, target, val_dataloader, val_iterator, maskidx, geometry, colorjitter, args)
        return patch

    def validate(self, i, patch, target, val_dataloader, val_iterator, maskidx, geometry, colorjitter, args):
        avg_CE = avg_L1 = 0.0
        val_num_sample = success = 0
        val_rel = 0.0
        asr6 = [0] * 6
        cont_pred = cont_gt = modified_images = None
        gripper = len(maskidx) == 1 and maskidx[0] == 6
        rb = ValReadback(self.val_batches, self.device)  # scalars + prediction maps stay on the device: ONE read-back behind the last batch
        orig_gt = []  # gripper mode: the unmodified labels of the kept samples (already on the host there: the filter has to look at them)
        with torch.no_grad():
            for _ in range(self.val_batches):
                data, val_iterator = next_or_restart(val_iterator, val_dataloader)
                pixel_values, labels, attention_mask, input_ids = to_dev(data, self.device)
                if gripper:  # keep only samples whose clean gripper prediction is right (TMA.py:222-248): the NEXT forward's batch depends
                    # on this forward's predictions — the one read-back per batch that cannot be deferred
                    clean = self.randomPatchTransform.im_process(pixel_values, mean=self.mean, std=self.std)
                    _, _, pre = self.model_loss(input_ids, attention_mask, clean, labels, ops.LOSS_CE, need_grad=False)
                    pm = pre.cpu().numpy()
                    gt = labels[:, 1:].cpu().numpy()
                    ok = [b for b in range(gt.shape[0]) if pm[b][gt[b] > 31743][-1] == gt[b][gt[b] > 31743][-1]]
                    if not ok:
                        print("No Correct in Val!")
                        continue
                    labels, attention_mask, input_ids = labels[ok], attention_mask[ok], input_ids[ok]
                    pixel_values = [pixel_values[b] for b in ok]
                    orig_gt.append(gt[ok])
                val_num_sample += labels.shape[0]
                modified_images = self._images(pixel_values, patch.detach(), geometry, colorjitter)
                newlabels = tma_target_labels(labels, target)
                _, scalars, pred = self.model_loss(input_ids, attention_mask, modified_images, newlabels, ops.LOSS_CE, need_grad=False)
                rb.add(scalars, pred, newlabels)
        host, maps = rb.read()
        n = max(len(maskidx), 1)
        for k, (sc, (p_np, gt_np)) in enumerate(zip(host, maps)):  # the reference's per-batch bookkeeping (TMA.py:250-290), in batch order
            cont_pred, cont_gt = self.decode_pred_gt_np(p_np, gt_np)
            val_rel += float(self.calculate_relative_distance_target(cont_pred, cont_gt))
            if gripper:
                tm = gt_np > 31743
                r = self.calculate_01_ASR(p_np[tm], orig_gt[k][tm])
                asr6 = [a + b for a, b in zip(asr6, r)]
            avg_L1 += float(torch.nn.functional.l1_loss(cont_pred, cont_gt)) if cont_pred.numel() else 0.0
            if cont_pred.numel():
                eq = (cont_pred.view(-1, n) == cont_gt.view(-1, n)).all(dim=1)
                success += int(eq.sum())
            avg_CE += float(np.float32(sc[1]))
        val_num_sample = max(val_num_sample, 1)
        avg_L1 /= val_num_sample
        avg_CE /= val_num_sample
        ASR = success / val_num_sample
        val_rel /= val_num_sample
        self.last_val_log = {"VAL_avg_CE_loss": avg_CE, "VAL_avg_L1_loss": avg_L1, "VAL_ASR": ASR, "VAL_inner_relatived_distance": val_rel}
        if len(maskidx) == 1 and maskidx[0] == 6:
            s02, n0, s12, n1, so2, no = asr6
            self.last_val_log.update({"ASR_02other": s02 / n0 if n0 else 0, "ASR_12other": s12 / n1 if n1 else 0,
                                      "ASR_other20": so2 / no if no else 0, "ALL_ASR_6": (s02 + s12) / (n0 + n1) if (n0 + n1) else 0})
        if wandb_enabled(args):
            wandb.log(self.last_val_log, step=i)
        dirs = []
        if avg_L1 < self.min_val_avg_L1_loss:
            self.min_val_avg_L1_loss = avg_L1
            dirs.append(self.save_patch(patch, f"{str(i)}"))
        dirs.append(self.save_patch(patch, "last"))
        for k, d in enumerate(dirs):
            if modified_images is not None and (k == 0 and len(dirs) == 2):
                path, _ = self.save_val_images(modified_images, d)
            else:
                path = os.path.join(d, "val_related_data")
                os.makedirs(path, exist_ok=True)
            if cont_pred is not None:
                torch.save(cont_pred.detach().cpu(), os.path.join(path, "continuous_actions_pred.pt"))
                torch.save(cont_gt.detach().cpu(), os.path.join(path, "continuous_actions_gt.pt"))
        self.val_CE_loss.append(avg_CE)
        self.val_L1_loss.append(avg_L1)
        self.val_ASR.append(success / val_num_sample)
        self.val_inner_relatived_distance.append(val_rel)
        self.save_info(self.save_dir)
        return val_iterator

    def save_info(self, path):
        """TMA.py:454-468."""
        self.dump_lists(["val_CE_loss", "val_L1_loss", "val_ASR", "val_inner_relatived_distance", "train_CE_loss",
                         "train_inner_avg_loss", "train_inner_relatived_distance"])
