"""UPA — untargeted position-aware attack. Mirrors VLAAttacker/white_patch/UPA.py:30-387.

reverse_direction=True (the CLI default): labels stay unmasked and the loss is
    alpha*mean_b(cos(e', l') + 1) + beta/(mean_b ||e' - l'||_2 + 1e-3)        (UPA.py:367-387)
over the soft-argmax of the first three action tokens (x, y, z). Other modes: `guide` (+CE against flipped targets,
UPA.py:130-131,143-144,358-364) and plain -CE (UPA.py:149-150). Every step clips the patch gradient to L1 norm 1e-3
before AdamW (UPA.py:157) — fused into K4. Validation: 100 batches, best patch by reverse-direction loss (UPA.py:193-275).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import ops
from ..labels import mask_labels as _mask_labels
from ..optim import CosineWarmupSchedule, PatchOptimizer
from .engine import AttackBase, ValReadback, next_or_restart, to_dev, wandb, wandb_enabled


class OpenVLAAttacker(AttackBase):
    val_batches = 100  # UPA.py:205

    def __init__(self, vla, processor=None, save_dir="", optimizer="pgd", resize_patch=False, alpha=0.8, belta=0.2):
        super().__init__(vla, processor, save_dir, optimizer, resize_patch)
        self.alpha, self.belta = alpha, belta
        self.avg_angle_loss, self.avg_distance_loss, self.avg_reserve_loss = [], [], []
        self.reverse_direction_loss = 100000
        self.min_val_avg_L1_loss = 1000000

    def mask_labels(self, labels, maskidx):
        return _mask_labels(labels, maskidx)

    def change_target(self, gt):
        """UPA.py:358-364, reproduced statement by statement. The reference assigns SEQUENTIALLY on the tensor it is rewriting: ties at
        31872 are first broken at random (one torch.randint draw, kept so the RNG stream matches), then every label > 31872 becomes
        31744, then every label < 31872 — which by now includes all the 31744s just written and the EOS token 2 — becomes 31999. The
        net effect of the shipped code is therefore "every non-ignored label -> 31999"; kept as is for parity (guide mode is off by
        default: UPA_wrapper.py passes guide=False)."""
        mask = gt != -100
        mid = mask & (gt == 31872)
        r = torch.randint(0, 2, gt[mid].shape, dtype=torch.bool).to(gt.device)
        gt[mid] = torch.where(r, torch.tensor(31744, dtype=gt.dtype, device=gt.device), torch.tensor(31999, dtype=gt.dtype, device=gt.device))
        gt[mask & (gt > 31872)] = 31744
        gt[mask & (gt < 31872)] = 31999
        return gt

    def _mode(self, guide, reverse_direction):
        if guide:
            return ops.LOSS_CE, 1.0
        if reverse_direction:
            return ops.LOSS_UPA, 1.0
        return ops.LOSS_CE, -1.0  # loss = -output.loss (UPA.py:150)

    def inner_step(self, patch, optimizer, pixel_values, input_ids, attention_mask, labels, geometry, mode, scale, scalars_out, k, do_step=True, read_scalars=True):
        """One iteration of the hot inner loop (UPA.py:127-159): [K0 per-image patch resize ->] K1 -> model -> K3 (K3h when the loss lives in the
        action slice) -> backward -> K2 / K2' (MULTI forms with resize_patch) -> K4 with the L1 clip (UPA.py:157) in front of AdamW."""
        pix = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch, mean=self.mean, std=self.std, geometry=geometry)
        # the reverse-direction loop never reads output.loss nor a full-vocabulary argmax (UPA.py:145-150,171-186): slice-only head (K3s)
        # (`read_scalars`: the loop prints / logs the loss terms of the LAST inner step of an outer iteration only, UPA.py:171-186: the others skip the fold)
        total, scalars, pred = self.model_loss(input_ids, attention_mask, pix, labels, mode, alpha=self.alpha, beta=self.belta, scale=scale, full_ce=False,
                                               read_scalars=read_scalars)
        total.backward()
        if do_step:
            stats = optimizer.step()  # K4: L1 clip 1e-3 -> AdamW -> clamp
            if read_scalars:
                scalars_out[k, 8:10] = stats
            optimizer.zero_grad()
        if read_scalars:
            scalars_out[k, :8] = scalars
        return pred

    def patchattack_unconstrained(self, train_dataloader, val_dataloader, num_iter=5000, target_action=np.zeros(7),
                                  patch_size=[3, 50, 50], lr=1 / 255, accumulate_steps=1, maskidx=[], warmup=20,
                                  filterGripTrainTo1=False, geometry=False, innerLoop=1, guide=False, reverse_direction=False, args=None):
        self.val_CE_loss, self.val_L1_loss, self.val_ASR, self.train_CE_loss, self.val_relative_distance = [], [], [], [], []
        dev = self.device
        patch = torch.rand(patch_size).to(dev)
        patch.requires_grad_(True)
        self.patch = patch
        adam = self.optimizer == "adamW"
        optimizer = PatchOptimizer(patch, lr, "adamW" if adam else "pgd", l1_clip=1e-3 if adam else 0.0)
        scheduler = CosineWarmupSchedule(optimizer, warmup, int(num_iter / accumulate_steps), 0.5) if adam else None
        train_iterator, val_iterator = iter(train_dataloader), iter(val_dataloader)
        scal = torch.zeros((max(innerLoop, 1), 10), dtype=torch.float32, device=dev)
        mode, scale = self._mode(guide, reverse_direction)

        for i in range(num_iter):
            data = next(train_iterator)
            if len(maskidx) == 1 and maskidx[0] == 6 and filterGripTrainTo1:
                labels, attention_mask, input_ids, pixel_values = self.filter_train(data)
            else:
                pixel_values, labels, attention_mask, input_ids = to_dev(data, dev)
            if not reverse_direction:
                labels = self.mask_labels(labels, maskidx)
            if guide:
                labels = self.change_target(labels)
            do_step = (i + 1) % accumulate_steps == 0 or (i + 1) == len(train_dataloader)
            every = os.environ.get("VAA_FULL_CE_EVERY_STEP", "0") == "1"  # (=1: every step folds its scalars; same patch bits)
            for inner_loop in range(innerLoop):
                self.inner_step(patch, optimizer, pixel_values, input_ids, attention_mask, labels, geometry, mode, scale, scal, inner_loop, do_step=do_step,
                                read_scalars=every or inner_loop == innerLoop - 1)
            if scheduler is not None and do_step:
                scheduler.step()
            host = scal[:innerLoop].cpu().numpy()
            self.assert_finite_state(patch, optimizer, host[:, :8], f"UPA outer iteration {i}")
            loss, angle_loss, distance_loss = float(host[-1, 0]), float(host[-1, 3]), float(host[-1, 4])
            print(f"loss: {loss}, " if reverse_direction else f"target_loss: {loss}")
            self.last_train_log = {"TRAIN_attack_loss(CE)": loss, "TRAIN_patch_gradient": float(host[-1, 9]),
                                   "TRAIN_LR": optimizer.param_groups[0]["lr"], "TRAIN_ANGLE_LOSS": angle_loss,
                                   "TRAIN_DISTANCE_LOSS": distance_loss}
            if wandb_enabled(args):
                wandb.log(self.last_train_log, step=i)
            self.train_CE_loss.append(loss)
            if i % 100 == 0:
                self.plot_loss()
                val_iterator = self.validate(i, patch, val_dataloader, val_iterator, maskidx, geometry, reverse_direction, mode, scale, args)
        return patch

    def validate(self, i, patch, val_dataloader, val_iterator, maskidx, geometry, reverse_direction, mode, scale, args):
        avg_angle = avg_dist = avg_res = 0.0
        val_num_sample = 0
        modified_images = None
        rb = ValReadback(self.val_batches, self.device)  # one read-back behind the last batch instead of one per batch
        with torch.no_grad():
            for _ in range(self.val_batches):
                data, val_iterator = next_or_restart(val_iterator, val_dataloader)
                pixel_values, labels, attention_mask, input_ids = to_dev(data, self.device)
                val_num_sample += labels.shape[0]
                modified_images = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch.detach(), mean=self.mean,
                                                                                      std=self.std, geometry=geometry)
                if not reverse_direction:
                    labels = self.mask_labels(labels, maskidx)
                _, scalars, _ = self.model_loss(input_ids, attention_mask, modified_images, labels, mode, alpha=self.alpha,
                                                beta=self.belta, scale=scale, need_grad=False, full_ce=False)
                rb.add(scalars)
        host, _ = rb.read()
        for s in host:
            avg_angle += float(np.float32(s[3]))
            avg_dist += float(np.float32(s[4]))
            avg_res += float(np.float32(s[0]))
        avg_angle /= val_num_sample
        avg_dist /= val_num_sample
        avg_res /= val_num_sample
        self.last_val_log = {"reverse_direction_loss": avg_res, "avg_angle_loss": avg_angle, "avg_distance_loss": avg_dist}
        if wandb_enabled(args):
            wandb.log(self.last_val_log, step=i)
        if avg_res < self.reverse_direction_loss:
            self.reverse_direction_loss = avg_res
            d = self.save_patch(patch, f"{str(i)}")
            self.save_val_images(modified_images, d)
        import os

        d = self.save_patch(patch, "last")
        os.makedirs(os.path.join(d, "val_related_data"), exist_ok=True)
        self.val_CE_loss.append(0)
        self.val_L1_loss.append(0)
        self.val_ASR.append(0 / val_num_sample)
        self.avg_angle_loss.append(avg_angle / val_num_sample)  # UPA.py:269-271 divides a second time
        self.avg_distance_loss.append(avg_dist / val_num_sample)
        self.avg_reserve_loss.append(avg_res / val_num_sample)
        self.save_info(self.save_dir)
        return val_iterator

    def save_info(self, path):
        """UPA.py:309-325 (file names differ from attribute names there)."""
        import os
        import pickle

        for fname, attr in (("val_relative_distance", "val_relative_distance"), ("val_CE_loss", "val_CE_loss"), ("val_L1_loss", "val_L1_loss"),
                            ("val_ASR", "val_ASR"), ("train_CE_loss", "train_CE_loss"), ("val_avg_angle_loss", "avg_angle_loss"),
                            ("val_avg_distance_loss", "avg_distance_loss"), ("val_avg_reserve_loss", "avg_reserve_loss")):
            with open(os.path.join(self.save_dir, f"{fname}.pkl"), "wb") as f:
                pickle.dump(getattr(self, attr), f)
