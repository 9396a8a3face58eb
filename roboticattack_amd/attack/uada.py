"""UADA — untargeted action-discrepancy attack, single GPU. Mirrors VLAAttacker/white_patch/UADA.py:33-418.

Same class name, constructor and `patchattack_unconstrained` signature, same RNG consumption order, same output
files (`<save_dir>/<iter>/patch.pt`, `last/patch.pt`, PNG dumps, pickled metric lists). The inner loop body is the
HIP path: K1 -> model -> K3 -> K2 -> K4, with no host synchronisation inside the innerLoop.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from ..labels import mask_labels as _mask_labels
from ..optim import CosineWarmupSchedule, PatchOptimizer
from .engine import AttackBase, ValReadback, next_or_restart, to_dev, wandb, wandb_enabled

IGNORE_INDEX = -100


class OpenVLAAttacker(AttackBase):
    val_batches = 1000  # UADA.py:202

    def __init__(self, vla, processor=None, save_dir="", optimizer="pgd", resize_patch=False):
        super().__init__(vla, processor, save_dir, optimizer, resize_patch)
        self.MSE_Distance_best = 10000  # UADA.py:59
        self.mse_weight = 5.0  # UADA.py:396
        self.loss_mode = ops.LOSS_UADA  # MSE + 1/CE (UADA.py:147)

    def mask_labels(self, labels, maskidx):
        return _mask_labels(labels, maskidx)

    # ------------------------------------------------------------------------------------------
    def inner_step(self, patch, optimizer, pixel_values, input_ids, attention_mask, labels, geometry, scalars_out, k):
        """One iteration of the hot inner loop (UADA.py:133-159). With K2' (a model that exposes its patch-embed weights) the step ends with ONE
        launch that adds K2''s partial tiles and applies AdamW + clamp; the logged gradient statistics are then folded once per outer iteration."""
        sink = self.fused_update_sink(optimizer)
        pix = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch, mean=self.mean, std=self.std, geometry=geometry, **({"grad_sink": sink} if sink is not None else {}))
        total, scalars, pred = self.model_loss(input_ids, attention_mask, pix, labels, self.loss_mode, w=self.mse_weight)
        total.backward()
        scalars_out[k, :8] = scalars
        if sink is not None and "partials" in sink:
            self.fused_update(sink, patch, optimizer, scalars)  # K2's final sum + K4 (AdamW + clamp(0,1)) in one launch
            self._stats_row = k
        else:
            scalars_out[k, 8:10] = optimizer.step()  # K4: AdamW + clamp(0,1); patch.data updated in place
            self._stats_row = None
        optimizer.zero_grad()
        return pred, pix

    def patchattack_unconstrained(self, train_dataloader, val_dataloader, num_iter=5000, target_action=np.zeros(7),
                                  patch_size=[3, 50, 50], lr=1 / 255, accumulate_steps=1, maskidx=[], warmup=20,
                                  filterGripTrainTo1=False, geometry=False, innerLoop=1, args=None):
        self.val_CE_loss, self.val_MSE_Distance, self.val_UAD = [], [], []
        self.train_CE_loss, self.train_MSE_distance_loss, self.train_UAD = [], [], []
        dev = self.device

        patch = torch.rand(patch_size).to(dev)  # CPU generator, then H2D: same values as UADA.py:104
        patch.requires_grad_(True)
        optimizer = PatchOptimizer(patch, lr, "adamW" if self.optimizer == "adamW" else "pgd")
        scheduler = None
        if self.optimizer == "adamW":
            scheduler = CosineWarmupSchedule(optimizer, warmup, int(num_iter / accumulate_steps), 0.5)
        train_iterator = iter(train_dataloader)
        val_iterator = iter(val_dataloader)
        scal = torch.zeros((max(innerLoop, 1), 10), dtype=torch.float32, device=dev)
        self.patch = patch

        for i in range(num_iter):
            train_relative_distance = {f"{idx}": [] for idx in maskidx}
            data = next(train_iterator)
            if len(maskidx) == 1 and maskidx[0] == 6 and filterGripTrainTo1:
                labels, attention_mask, input_ids, pixel_values = self.filter_train(data)
            else:
                pixel_values, labels, attention_mask, input_ids = to_dev(data, dev)
            labels = self.mask_labels(labels, maskidx)

            pred = None
            for inner_loop in range(innerLoop):
                pred, _ = self.inner_step(patch, optimizer, pixel_values, input_ids, attention_mask, labels, geometry, scal, inner_loop)

            if scheduler is not None and ((i + 1) % accumulate_steps == 0 or (i + 1) == len(train_dataloader)):
                scheduler.step()

            if getattr(self, "_stats_row", None) is not None:  # fused update: the last step's {sum|g|, mean g} from its block sums
                scal[self._stats_row, 8:10] = optimizer.last_stats
            # one read-back per outer iteration (the reference syncs 4-5 times per inner step)
            host = scal[:innerLoop].cpu().numpy()
            self.assert_finite_state(patch, optimizer, host[:, :8], f"UADA outer iteration {i}")
            self.train_CE_loss.extend(host[:, 1].tolist())
            self.train_MSE_distance_loss.extend(host[:, 0].tolist())
            self.train_UAD.extend(host[:, 7].tolist())
            celoss, total, UAD, log_patch_grad = float(host[-1, 1]), float(host[-1, 0]), float(host[-1, 7]), float(host[-1, 9])
            cont_pred, cont_gt = self.decode_pred_gt(pred, labels)
            train_relative_distance = self.calculate_relative_distance(cont_pred, cont_gt, maskidx, train_relative_distance)
            train_logdata = {"TRAIN_attack_loss(CE)": celoss, "TRAIN_patch_gradient": log_patch_grad,
                             "TRAIN_LR": optimizer.param_groups[0]["lr"], "TRAIN_attack_loss (MSE_Distance)": total, "TRAIN_UAD": UAD}
            for key, value in train_relative_distance.items():
                train_logdata[f"train_rd_{key}"] = sum(value) / len(value)
            self.last_train_log = train_logdata
            if wandb_enabled(args):
                wandb.log(train_logdata, step=i)

            if i % 100 == 0:
                self.plot_loss()
                val_iterator = self.validate(i, patch, val_dataloader, val_iterator, maskidx, geometry, args)
        return patch

    # ------------------------------------------------------------------------------------------
    def validate(self, i, patch, val_dataloader, val_iterator, maskidx, geometry, args):
        """UADA.py:193-292: no-grad sweep, best-patch selection by summed MSE distance / #samples, `last/` always."""
        avg_CE_loss = avg_MSE_Distance = avg_UAD = 0.0
        val_num_sample = 0
        relative_distance = {f"{idx}": [] for idx in maskidx}
        modified_images = None
        val_UAD = 0.0
        rb = ValReadback(self.val_batches, self.device)  # nothing is read back inside the loop: the host enqueues batch after batch
        with torch.no_grad():
            for _ in range(self.val_batches):
                data, val_iterator = next_or_restart(val_iterator, val_dataloader)
                pixel_values, labels, attention_mask, input_ids = to_dev(data, self.device)
                val_num_sample += labels.shape[0]
                modified_images = self.randomPatchTransform.apply_random_patch_batch(
                    pixel_values, patch.detach(), mean=self.mean, std=self.std, geometry=geometry)
                labels = self.mask_labels(labels, maskidx)
                _, scalars, pred = self.model_loss(input_ids, attention_mask, modified_images, labels, ops.LOSS_UADA_DDP,
                                                   w=self.mse_weight, need_grad=False)
                rb.add(scalars, pred, labels)
        host, maps = rb.read()
        for s, (p_np, gt_np) in zip(host, maps):  # the reference's per-batch bookkeeping (UADA.py:229-246), in batch order
            cont_pred, cont_gt = self.decode_pred_gt_np(p_np, gt_np)
            relative_distance = self.calculate_relative_distance(cont_pred, cont_gt, maskidx, relative_distance)
            avg_MSE_Distance += float(np.float32(s[2]))
            val_UAD = float(np.float32(s[7]))
            avg_UAD += val_UAD
            avg_CE_loss += float(np.float32(s[1]))
        avg_MSE_Distance /= val_num_sample
        avg_UAD /= val_num_sample
        avg_CE_loss /= val_num_sample
        log_data = {"VAL_MSE_Distance": avg_MSE_Distance, "VAL_UAD": val_UAD}
        for key, value in relative_distance.items():
            log_data[f"val_rd_{key}"] = sum(value) / len(value)
        self.last_val_log = log_data
        if wandb_enabled(args):
            wandb.log(log_data, step=i)
        if avg_MSE_Distance < self.MSE_Distance_best:
            self.MSE_Distance_best = avg_MSE_Distance
            d = self.save_patch(patch, f"{str(i)}")
            _, pil = self.save_val_images(modified_images, d)
            if wandb_enabled(args):
                wandb.log({"AdvImg": [wandb.Image(p) for p in pil]})
        d = self.save_patch(patch, "last")
        _, pil = self.save_val_images(modified_images, d)
        if wandb_enabled(args):
            wandb.log({"Last_Step_AdvImg": [wandb.Image(p) for p in pil]})
        self.val_CE_loss.append(avg_CE_loss)
        self.val_MSE_Distance.append(avg_MSE_Distance)
        self.val_UAD.append(avg_UAD)
        self.save_info(path=self.save_dir)
        return val_iterator

    def save_info(self, path):
        """UADA.py:341-353."""
        self.dump_lists(["train_CE_loss", "train_MSE_distance_loss", "train_UAD", "val_CE_loss", "val_MSE_Distance", "val_UAD"])

    # kept for API parity with the reference class (UADA.py:381-418); the loop itself uses the fused K3
    def weighted_loss(self, logits, labels, maskid=None):
        total, scalars, _ = ops.DiscrepancyLoss.apply(logits.contiguous(), labels, ops.LOSS_UADA_DDP, self.mse_weight, 0.8, 0.2, 1.0,
                                                      ops.LAYOUT_FULL)
        return total, scalars[7]
