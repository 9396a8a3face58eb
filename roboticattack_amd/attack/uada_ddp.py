"""UADA, data-parallel over the GPUs of one node. Mirrors VLAAttacker/white_patch/UADA_ddp.py:36-344.

One process per GPU under torchrun (RANK / WORLD_SIZE / LOCAL_RANK), `torch.distributed` backend "nccl" (= RCCL over
xGMI). Same constructor arguments and `_attack_entry(rank, params, world)` / `attack(rank, world)` entry points as the
reference. What differs by design:
  * K1/K2 run on the GPU (the reference binds the transform to the CPU while the model is still there, Appendix A-D9);
  * no DistributedDataParallel wrapper around 7.5 B frozen parameters: the single trainable tensor is synchronised by
    one fused all-reduce per inner step (`dist.PatchGradSync`) and DDP's mean is folded into K4's grad_scale;
  * every rank seeds Python/NumPy with 42 (UADA_wrapper_ddp.py:53), so transform parameters per local index are
    identical across ranks, exactly like the reference.
`bs` is the PER-RANK batch (UADA_ddp.py:158); data are sharded by rank (tf.data shard in the reference, :157-160).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib
from .. import dist as vdist
from .. import ops
from ..labels import mask_labels as _mask_labels
from ..optim import CosineWarmupSchedule, PatchOptimizer
from .engine import AttackBase, ValReadback, to_dev, wandb, wandb_enabled


def default_model_factory(vla_path: str, device):
    """`vla_path` forms: "random:openvla-7b" / "random:tiny" (random init), "surrogate[:seed]", or a local HF checkpoint dir."""
    from ..openvla_model import build_openvla, load_hf_openvla, openvla_7b_cfg, tiny_cfg

    if vla_path.startswith("surrogate"):
        from ..surrogate import SurrogateVLA

        seed = int(vla_path.split(":")[1]) if ":" in vla_path else 0
        return SurrogateVLA(seed=seed).to(device)
    if vla_path.startswith("random:"):
        cfg = tiny_cfg() if vla_path.endswith("tiny") else openvla_7b_cfg()
        return build_openvla(cfg, device=device, dtype=torch.bfloat16)  # the transform hands over bf16 pixel_values, as the reference does
    if os.path.isdir(vla_path):
        return load_hf_openvla(build_openvla(openvla_7b_cfg(), device=device), vla_path)
    raise FileNotFoundError(
        f"no local OpenVLA checkpoint at {vla_path!r} (this image has no network); use 'random:openvla-7b' for shape-exact runs")


def default_dataset_factory(dataset_name: str, bs: int, rank: int, world: int):
    """Synthetic BridgeData-shaped loaders, sharded by rank through the seed (stands in for RLDSDataset.shard)."""
    from ..synthetic import SyntheticLoader

    return (SyntheticLoader(bs, seed=1234 + 1000003 * rank, kind="noise"), SyntheticLoader(bs, seed=99991 + 1000003 * rank, kind="noise"))


class OpenVLAAttacker(AttackBase):
    val_batches = 100  # UADA_ddp.py:240
    val_every = 200  # UADA_ddp.py:233

    def __init__(self, vla_path, dataset_name, save_dir="", resize_patch=False, patch_size=[3, 50, 50], lr=0.01, bs=1, warmup=20,
                 num_iter=10000, maskidx=[], innerLoop=1, geometry=True, use_wandb=True, MSE_weights=1,
                 model_factory=None, dataset_factory=None, device=None, attack_type="UADA", alpha=0.8, belta=0.2, target_action=0.0):
        """`attack_type`, `alpha`, `belta`, `target_action` are EXTENSIONS (the reference ships DDP for UADA only):
        "UPA" = UPA.py's reverse-direction loss + L1 grad clip, "TMA" = TMA.py's target-token CE, same data-parallel loop."""
        rank, world, local = vdist.env_rank_world()
        if device is None:
            device = vdist.local_device()
        self._rank, self._world = rank, world
        vla = (model_factory or default_model_factory)(vla_path, device)
        super().__init__(vla, None, save_dir, "adamW", resize_patch)
        self.device = torch.device(device)
        self.train_loader, self.val_loader = (dataset_factory or default_dataset_factory)(dataset_name, bs, rank, world)
        self.MSE_Distance_best = 1000000
        self.bs, self.lr, self.warmup, self.num_iter = bs, lr, warmup, num_iter
        self.maskidx, self.innerLoop, self.geometry = maskidx, innerLoop, geometry
        self.use_wandb = bool(use_wandb) and use_wandb != "false"  # Appendix A-D14
        self.patch_size = patch_size
        self.val_CE_loss, self.val_MSE_Distance, self.val_UAD = [], [], []
        self.MSE_weights = MSE_weights
        if attack_type not in ("UADA", "UPA", "TMA"):
            raise ValueError(f"attack_type must be UADA, UPA or TMA, got {attack_type!r}")
        self.attack_type, self.alpha, self.belta, self.target_action = attack_type, alpha, belta, target_action

    def setup(self, rank, world_size):
        vdist.init_process_group(device=self.device if self.device.type == "cuda" else None)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)

    def cleanup(self):
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()

    def mask_labels(self, labels, maskidx):
        return _mask_labels(labels, maskidx)

    def attack(self, rank, world_size):
        self.setup(rank, world_size)
        try:
            return self._attack(rank, world_size)
        finally:  # also when the run ends with NonFiniteAttackState: a process that leaves its group open can block at exit
            self.cleanup()

    def _attack(self, rank, world_size):
        dev = self.device
        if rank == 0:
            patch = torch.rand(self.patch_size).to(dev)  # UADA_ddp.py:140-141
        else:
            patch = torch.empty(self.patch_size).to(dev)
        vdist.broadcast_patch(patch, src=0)  # C1
        patch.requires_grad_(True)
        self.patch = patch
        optimizer = PatchOptimizer(patch, self.lr, "adamW", l1_clip=1e-3 if self.attack_type == "UPA" else 0.0)  # UPA.py:157
        scheduler = CosineWarmupSchedule(optimizer, self.warmup, int(self.num_iter), 0.5)
        if self.attack_type == "TMA":
            from ..labels import tma_target_tokens

            self._tma_target = tma_target_tokens(float(self.target_action) * torch.ones(7).numpy(), self.maskidx, self.action_tokenizer).to(dev)
        sync = vdist.PatchGradSync(patch.numel(), 4, dev)
        pick = torch.tensor([1, 2, 7, 0], dtype=torch.int64, device=dev)  # CE, w^2*MSE, UAD, total of K3's scalars
        inv_world = 1.0 / world_size
        # UADA on a model that exposes its patch-embed weights: K2's final sum, K3's fold and the message packing are ONE launch
        fused = self.attack_type == "UADA" and self.fused_ddp_available()
        scalars = torch.zeros(8, dtype=torch.float32, device=dev)

        for i, data in enumerate(self.train_loader):
            if i == self.num_iter:
                break
            pixel_values, labels, attention_mask, input_ids = to_dev(data, dev)
            labels = self._prepare_labels(labels)
            s_sum = sync.buf[sync.n_grad :]
            device_failure = None  # a library call of THIS rank reported a device-side failure (the sticky word, e.g. a hand-over that timed out)
            for inner_loop in range(self.innerLoop):
                exchanged = False
                if device_failure is None:
                    try:
                        optimizer.zero_grad()
                        # the loop reads the loss scalars of the LAST inner step only (UADA_ddp.py:214-221, below): the full-vocabulary CE is evaluated there
                        full_ce = inner_loop == self.innerLoop - 1 or os.environ.get("VAA_FULL_CE_EVERY_STEP", "0") == "1"  # (=1: K3h behind K3s on every step; same patch bits)
                        if fused and world_size == 1:  # nothing to exchange: K4 runs inside the epilogue launch (five launches per step)
                            self.fused_ddp_step(pixel_values, patch, input_ids, attention_mask, labels, self.geometry, float(self.MSE_weights),
                                                sync.buf, scalars, optimizer=optimizer, full_ce=full_ce)
                            s_sum = sync.buf[sync.n_grad :]
                            continue
                        if fused:
                            self.fused_ddp_step(pixel_values, patch, input_ids, attention_mask, labels, self.geometry, float(self.MSE_weights),
                                                sync.buf, scalars, full_ce=full_ce)
                            exchanged = True
                            g_sum, s_sum = sync.allreduce_packed()  # C3 + C4 in one message: [grad | CE, MSE, UAD, total]
                        else:
                            pix = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch, mean=self.mean, std=self.std,
                                                                                      geometry=self.geometry)
                            total, scalars, pred = self.model_loss(input_ids, attention_mask, pix, labels, self._loss_mode(), w=float(self.MSE_weights),
                                                                   alpha=self.alpha, beta=self.belta, full_ce=full_ce, read_scalars=full_ce)
                            total.backward()  # K2 inside
                            exchanged = True
                            g_sum, s_sum = sync.allreduce_step(patch.grad, scalars, pick)
                        optimizer.step(grad=g_sum.view_as(patch), grad_scale=inv_world)  # K4, DDP mean folded in
                    except _lib.VaaError as e:
                        # The failure word is sticky: every library call of this rank fails until it is polled. Raising here would leave the OTHER
                        # ranks waiting in this step's all-reduce until the RCCL timeout: poll it now (the poll clears it), keep the
                        # collective's cadence with a NaN message for the rest of the outer iteration, and let the all-reduced verdict below
                        # take every rank out of the loop together.
                        device_failure = str(e)
                        try:
                            ops.async_error_check()
                        except _lib.VaaError:
                            pass
                if device_failure is not None and world_size > 1 and not exchanged:
                    sync.buf.fill_(float("nan"))
                    _, s_sum = sync.allreduce_packed()
            scheduler.step()
            s = (s_sum * inv_world).cpu().numpy()
            self.assert_finite_state(patch, optimizer, s, f"{self.attack_type} (data parallel) outer iteration {i}", all_ranks=True, device_failure=device_failure)
            # UADA_ddp.py:207,216-217: `patch.grad.mean()` AFTER DistributedDataParallel averaged the gradient, i.e. the mean of the averaged
            # gradient (identical on every rank, so the reference's MAX all-reduce of it is the value itself): K4 reports it
            log_patch_grad = float(optimizer.last_stats[1].item())
            train_logdata = {"TRAIN_attack_loss(CE)": float(s[0]), "TRAIN_patch_gradient": log_patch_grad,
                             "TRAIN_LR": optimizer.param_groups[0]["lr"], "TRAIN_attack_loss (MSE_Distance)": float(s[1]),
                             "TRAIN_UAD": float(s[2])}
            self.last_train_log = train_logdata
            if rank == 0 and self.use_wandb and wandb is not None:
                wandb.log(train_logdata, step=i)
            if i % self.val_every == 0:
                self.validate(i, patch, rank)
        return patch

    def _loss_mode(self):
        return {"UADA": ops.LOSS_UADA_DDP, "UPA": ops.LOSS_UPA, "TMA": ops.LOSS_CE}[self.attack_type]

    def _prepare_labels(self, labels):
        if self.attack_type == "UADA":
            return self.mask_labels(labels, self.maskidx)
        if self.attack_type == "TMA":
            from ..labels import tma_target_labels

            return tma_target_labels(labels, self._tma_target)
        return labels  # UPA reverse_direction: labels stay unmasked (UPA.py:127-129)

    def validate(self, i, patch, rank):
        """UADA_ddp.py:233-324: 100 local val batches, 3 scalar all-reduces (C5), rank 0 writes the files."""
        avg_CE = avg_MSE = avg_UAD = 0.0
        modified_images = None
        rb = ValReadback(self.val_batches, self.device)  # one read-back behind the last batch instead of one per batch
        with torch.no_grad():
            for j, data in enumerate(self.val_loader):
                if j == self.val_batches:
                    break
                pixel_values, labels, attention_mask, input_ids = to_dev(data, self.device)
                modified_images = self.randomPatchTransform.apply_random_patch_batch(pixel_values, patch.detach(), mean=self.mean,
                                                                                      std=self.std, geometry=self.geometry)
                labels = self._prepare_labels(labels)
                _, scalars, _ = self.model_loss(input_ids, attention_mask, modified_images, labels, self._loss_mode(),
                                                w=float(self.MSE_weights), alpha=self.alpha, beta=self.belta, need_grad=False)
                rb.add(scalars)
        host, _ = rb.read()
        for s in host:
            avg_MSE += float(np.float32(s[2] if self.attack_type == "UADA" else s[0]))  # selection metric: MSE distance (UADA) or the attack loss
            avg_UAD += float(np.float32(s[7]))
            avg_CE += float(np.float32(s[1]))
        avg_MSE /= self.val_batches
        avg_UAD /= self.val_batches
        avg_CE /= self.val_batches
        g_MSE = vdist.allreduce_scalar(avg_MSE, "AVG", self.device)
        g_UAD = vdist.allreduce_scalar(avg_UAD, "AVG", self.device)
        g_CE = vdist.allreduce_scalar(avg_CE, "AVG", self.device)
        self.last_val_log = {"VAL_MSE_Distance": g_MSE, "VAL_UAD": g_UAD}
        if rank == 0:
            if g_MSE < self.MSE_Distance_best:
                self.MSE_Distance_best = g_MSE
                d = self.save_patch(patch, f"{str(i)}")
                _, pil = self.save_val_images(modified_images, d)
                if self.use_wandb and wandb is not None:
                    wandb.log(self.last_val_log, step=i)
                    wandb.log({"AdvImg": [wandb.Image(p) for p in pil]})
            d = self.save_patch(patch, "last")
            os.makedirs(os.path.join(d, "val_related_data"), exist_ok=True)
            self.val_CE_loss.append(g_CE)
            self.val_MSE_Distance.append(g_MSE)
            self.val_UAD.append(g_UAD)

    @classmethod
    def run(cls, **instance_params):
        """UADA_ddp.py:326-336 spawns one process per visible GPU; torchrun is the supported launcher here."""
        import torch.multiprocessing as mp

        world_size = max(torch.cuda.device_count(), 1)
        mp.spawn(cls._spawn_entry, args=(instance_params, world_size), nprocs=world_size)

    @staticmethod
    def _spawn_entry(rank, instance_params, world_size):
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size))
        OpenVLAAttacker._attack_entry(rank, instance_params, world_size)

    @staticmethod
    def _attack_entry(rank, instance_params, world_size):
        """UADA_ddp.py:338-344."""
        instance = OpenVLAAttacker(**instance_params)
        return instance.attack(rank, world_size)
