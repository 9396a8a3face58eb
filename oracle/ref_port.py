"""ORACLE (test infrastructure, NOT product code): PyTorch-CPU restatement of the reference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file. The
product path (roboticattack_amd/*) never does; it fails loudly when libvaa_hip.so is missing.

Each function restates, op for op, what the reference executes on CPU, citing file:line under
/root/reference (read-only, absent on the GPU box). Pinning: every function here is checked in
tests/test_oracle_golden.py against vectors recorded by tools/gen_golden.py, which ran the
reference's own functions in the survey container (tests/golden/*.npz).

Third-party arithmetic that is NOT in the reference tree:
  * transformers==4.40.1 `AdamW`  -> `HFAdamW` below: restated from the published algorithm,
    **parity unpinned** (the class is absent from the installed transformers 5.x and the
    reference holds no known-answer test for it).
  * transformers `get_cosine_schedule_with_warmup` -> `cosine_lambda`; pinned against the
    installed transformers (tests/golden/sched.npz).
  * HF Llama causal-LM loss -> `hf_ce`; restated, pinned only through torch's F.cross_entropy.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_INDEX = -100
MEAN = [torch.tensor([0.484375, 0.455078125, 0.40625]), torch.tensor([0.5, 0.5, 0.5])]  # UADA.py:56
STD = [torch.tensor([0.228515625, 0.2236328125, 0.224609375]), torch.tensor([0.5, 0.5, 0.5])]  # UADA.py:57


# --------------------------------------------------------------------------------------------
# a-2  random geometry (appply_random_transform.py:26-41, 80-91)
# --------------------------------------------------------------------------------------------
def rotation_matrix(theta_deg: float) -> np.ndarray:
    """appply_random_transform.py:26-34 — float64 cos/sin cast to a float32 3x3."""
    t = np.deg2rad(theta_deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)


def shear_matrix(shx: float, shy: float) -> np.ndarray:
    """appply_random_transform.py:36-41."""
    return np.array([[1, shx, 0], [shy, 1, 0], [0, 0, 1]], dtype=np.float32)


def combined_transform_matrix(max_angle=30, max_shx=0.2, max_shy=0.2) -> np.ndarray:
    """appply_random_transform.py:80-91 — p=0.2 identity, else S(shx,shy) @ R(angle) in float32."""
    if np.random.rand() < 0.2:
        return np.eye(3, dtype=np.float32)
    angle = np.random.uniform(-max_angle, max_angle)
    shx = np.random.uniform(-max_shx, max_shx)
    shy = np.random.uniform(-max_shy, max_shy)
    return np.dot(shear_matrix(shx, shy), rotation_matrix(angle))


def draw_params(batch: int, ph: int, pw: int, geometry: bool, img: int = 224):
    """RNG draw order of apply_random_patch_batch (appply_random_transform.py:120-128), per image:
    random.randint(x), random.randint(y), then (geometry only) np.random.rand [, uniform x3].
    Returns xy int32 [B,2] (x,y) and theta float32 [B,2,3] (identity rows when geometry is False)."""
    xy = np.zeros((batch, 2), dtype=np.int32)
    theta = np.zeros((batch, 2, 3), dtype=np.float32)
    for b in range(batch):
        x = random.randint(0, img - pw)
        y = random.randint(0, img - ph)
        xy[b] = (x, y)
        m = combined_transform_matrix() if geometry else np.eye(3, dtype=np.float32)
        theta[b] = m[:2, :]
    return xy, theta


# --------------------------------------------------------------------------------------------
# a-1 / a-3  K1 forward (and K2 through autograd)
# --------------------------------------------------------------------------------------------
def _normalize(images, mean, std):
    """appply_random_transform.py:16-19."""
    images = images - mean[None, :, None, None]
    images = images / std[None, :, None, None]
    return images


def to_tensor_u8(img_hwc_u8: np.ndarray) -> torch.Tensor:
    """torchvision ToTensor on a PIL RGB image [3p]: u8 HWC -> f32 CHW, true division by 255."""
    t = torch.from_numpy(np.ascontiguousarray(img_hwc_u8)).permute(2, 0, 1).contiguous()
    return t.to(torch.float32).div(255)


def apply_affine_transform(canvas: torch.Tensor, theta23: torch.Tensor) -> torch.Tensor:
    """appply_random_transform.py:93-102 — affine_grid + grid_sample(bilinear, border, align_corners=False)."""
    grid = F.affine_grid(theta23[None], [1, *canvas.shape], align_corners=False)
    return F.grid_sample(canvas[None], grid, align_corners=False, padding_mode="border")


def apply_random_patch_batch(images_u8, patch, xy, theta, geometry: bool, threshold_mode: str = "lt-20", return_keep=False):
    """appply_random_transform.py:104-136 with the random draws hoisted out (xy/theta from `draw_params`).

    threshold_mode "lt-20" is `torch.where(canvas < -20, im, canvas)` (:131); "ne-100" is the
    `paste_patch_fix` / `random_paste_patch` rule `torch.where(canvas != -100, canvas, im)` (:153,:179).
    Output f32 [B,6,224,224]; differentiable w.r.t. `patch` (that backward IS kernel K2's definition)."""
    outs, keeps = [], []
    ph, pw = patch.shape[1], patch.shape[2]
    for b in range(len(images_u8)):
        im = to_tensor_u8(np.asarray(images_u8[b]))  # :108
        canvas = torch.ones(3, im.shape[1], im.shape[2]) * -100  # :111
        x, y = int(xy[b][0]), int(xy[b][1])
        canvas[:, y : y + ph, x : x + pw] = patch  # :125
        if geometry:
            canvas = apply_affine_transform(canvas, torch.as_tensor(theta[b]))  # :128-129 -> [1,3,H,W]
        if threshold_mode == "lt-20":
            cond = canvas < -20
            im = torch.where(cond, im, canvas)  # :131
            keep = ~cond
        else:
            keep = canvas != -100
            im = torch.where(keep, canvas, im)  # :179
        im0 = _normalize(im, MEAN[0], STD[0])  # :132
        im1 = _normalize(im, MEAN[1], STD[1])  # :133
        outs.append(torch.cat([im0, im1], dim=1))  # :135
        keeps.append(keep.reshape(1, 3, im.shape[-2], im.shape[-1]))
    out = torch.cat(outs, dim=0)  # :136
    if return_keep:
        return out, torch.cat(keeps, dim=0)
    return out


def patch_grad_via_autograd(images_u8, patch, xy, theta, geometry, gout_bf16, threshold_mode="lt-20"):
    """K2's definition: d<out.to(bf16), g>/d patch by autograd, as UADA.py:142-148 does implicitly."""
    p = patch.detach().clone().requires_grad_(True)
    out = apply_random_patch_batch(images_u8, p, xy, theta, geometry, threshold_mode)
    out.to(torch.bfloat16).backward(gradient=gout_bf16)
    return p.grad.detach()


# --------------------------------------------------------------------------------------------
# a-1 with resize_patch=True (BASELINE config 5): appply_random_transform.py:113-118, Appendix A-D2 semantics
# --------------------------------------------------------------------------------------------
def resize_patch(patch: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """`transforms.Resize((h, w))(patch)` (:116) [3p torchvision 0.17 on a tensor: bilinear, antialias=True]."""
    return F.interpolate(patch[None], size=(h, w), mode="bilinear", antialias=True, align_corners=False)[0]


def draw_params_resized(batch: int, ph: int, pw: int, geometry: bool, img: int = 224):
    """Draw order per image with resize_patch=True (:114, :123-128): random.uniform(0.61, 1.39) (scale), then randint(x),
    randint(y) against the RESIZED size, then (geometry) the transform matrix. Returns sizes [B,2] (h,w), xy, theta."""
    sizes = np.zeros((batch, 2), dtype=np.int32)
    xy = np.zeros((batch, 2), dtype=np.int32)
    theta = np.zeros((batch, 2, 3), dtype=np.float32)
    for b in range(batch):
        scale = random.uniform(0.61, 1.39)  # :114
        h, w = int(ph * scale), int(pw * scale)  # :115 (base patch size: A-D2)
        sizes[b] = (h, w)
        xy[b] = (random.randint(0, img - w), random.randint(0, img - h))
        m = combined_transform_matrix() if geometry else np.eye(3, dtype=np.float32)
        theta[b] = m[:2, :]
    return sizes, xy, theta


def apply_random_patch_batch_resized(images_u8, patch, sizes, xy, theta, geometry: bool, return_keep=False):
    """:104-136 with resize_patch=True and the draws hoisted out: image b pastes resize(base patch, sizes[b])."""
    outs, keeps = [], []
    for b in range(len(images_u8)):
        pb = resize_patch(patch, int(sizes[b][0]), int(sizes[b][1]))
        o = apply_random_patch_batch(images_u8[b : b + 1], pb, xy[b : b + 1], theta[b : b + 1], geometry, "lt-20", return_keep=return_keep)
        if return_keep:
            outs.append(o[0])
            keeps.append(o[1])
        else:
            outs.append(o)
    out = torch.cat(outs, dim=0)
    return (out, torch.cat(keeps, dim=0)) if return_keep else out


def patch_grad_resized_via_autograd(images_u8, patch, sizes, xy, theta, geometry, gout_bf16):
    p = patch.detach().clone().requires_grad_(True)
    out = apply_random_patch_batch_resized(images_u8, p, sizes, xy, theta, geometry)
    out.to(torch.bfloat16).backward(gradient=gout_bf16)
    return p.grad.detach()


# --------------------------------------------------------------------------------------------
# a-6  label masking
# --------------------------------------------------------------------------------------------
def mask_labels(labels: torch.Tensor, maskidx) -> torch.Tensor:
    """UADA.py:371-379 / UADA_ddp.py:89-97 (in place)."""
    mask = labels > 31743
    masked = labels[mask]
    masked = masked.view(masked.shape[0] // 7, 7)
    template = torch.ones_like(masked) * -100
    for idx in maskidx:
        template[:, idx] = masked[:, idx]
    labels[labels > 2] = template.view(-1)
    return labels


def tma_target_labels(labels: torch.Tensor, target_tokens: torch.Tensor) -> torch.Tensor:
    """TMA.py:124-129 — every non-ignored label position of a row is overwritten by the 8-entry target vector
    (7 target action tokens with non-maskidx entries set to -100, then EOS; TMA.py:93-99)."""
    new = []
    for j in range(labels.shape[0]):
        t = labels[j].clone()
        t[t != -100] = target_tokens
        new.append(t.unsqueeze(0))
    return torch.cat(new, dim=0)


# --------------------------------------------------------------------------------------------
# a-7c  action de-tokenisation + UAD metric
# --------------------------------------------------------------------------------------------
BINS = np.linspace(-1, 1, 256)  # action_tokenizer.py:31
BIN_CENTERS = (BINS[:-1] + BINS[1:]) / 2.0  # action_tokenizer.py:32


def decode_token_ids_to_actions(token_ids: np.ndarray) -> np.ndarray:
    """action_tokenizer.py:49-68."""
    d = 32000 - token_ids
    d = np.clip(d - 1, a_min=0, a_max=BIN_CENTERS.shape[0] - 1)
    return BIN_CENTERS[d]


def cal_uad(pred_tokens: torch.Tensor, gt_tokens: torch.Tensor) -> torch.Tensor:
    """UADA.py:408-418."""
    gt = torch.tensor(decode_token_ids_to_actions(gt_tokens.clone().detach().cpu().numpy()))
    pr = torch.tensor(decode_token_ids_to_actions(pred_tokens.clone().detach().cpu().numpy()))
    max_distance = torch.where(gt > 0, torch.abs(gt - (-1)), torch.abs(gt - 1))
    return (torch.abs(pr - gt) / max_distance).mean()


# --------------------------------------------------------------------------------------------
# a-7 / a-7b / a-7d / a-7e  losses
# --------------------------------------------------------------------------------------------
def hf_ce(logits: torch.Tensor, labels: torch.Tensor, n_img: int = 256) -> torch.Tensor:
    """HF Llama `.loss` as reached through modeling_prismatic.py:395-415 [3p transformers 4.40.1]:
    multimodal labels = [labels[:, :1], -100 x 256, labels[:, 1:]]; shift; fp32 mean CE, ignore -100."""
    B = labels.shape[0]
    mm = torch.cat([labels[:, :1], torch.full((B, n_img), -100, dtype=labels.dtype), labels[:, 1:]], dim=1)
    sl = logits[:, :-1, :].float().contiguous()
    tl = mm[:, 1:].contiguous()
    return F.cross_entropy(sl.view(-1, sl.shape[-1]), tl.view(-1), ignore_index=-100)


def uada_weighted_loss(logits: torch.Tensor, labels: torch.Tensor, mse_weight: float = 5.0):
    """UADA.py:381-406 (weight 5) / UADA_ddp.py:99-124 (weight = MSE_weights). Returns (loss, UAD)."""
    temp_label = labels[:, 1:]
    action_mask = temp_label > 2
    temp_logits = logits[:, :, 31744:32000]
    action_logits = temp_logits[:, -temp_label.shape[-1] - 1 : -1, :]
    action_logits = action_logits[action_mask]
    reweigh = torch.arange(1, 257) / 256
    prob = F.softmax(action_logits, dim=-1)
    r = (prob * reweigh).sum(dim=-1)
    hard = temp_label[action_mask]  # boolean indexing -> a copy, int64
    hard[hard > 31872] = 31999
    hard[hard <= 31872] = 31744
    hard[hard == 31999] = 1 / 256  # int64 assignment truncates 1/256 to 0 (Appendix A-D10) — reproduced
    hard[hard == 31744] = 1
    uad = cal_uad(action_logits.argmax(dim=-1) + 31744, temp_label[action_mask])
    loss = F.mse_loss(mse_weight * r.contiguous(), mse_weight * hard.float().contiguous())
    return loss, uad


def upa_weighted_loss(logits: torch.Tensor, labels: torch.Tensor, alpha: float, belta: float, n_img: int = 256):
    """UPA.py:367-387. Returns (total, angle_loss, distance_loss)."""
    temp_label = labels[:, 1:]
    action_mask = temp_label != -100
    temp_logits = logits[:, :, 31744:32000]
    action_logits = temp_logits[:, n_img:-1]
    reweigh = torch.arange(1, 257)
    prob = F.softmax(action_logits, dim=-1)
    e = (prob * reweigh).sum(dim=-1)
    xyz_e = torch.cat([row[action_mask[i]].unsqueeze(0) for i, row in enumerate(e)], dim=0)[:, :3]
    xyz_l = (torch.cat([row[action_mask[i]].unsqueeze(0) for i, row in enumerate(temp_label)], dim=0) - 31743)[:, :3]
    xyz_e = (xyz_e - 1) / 255
    xyz_l = (xyz_l - 1) / 255
    cos = F.cosine_similarity(xyz_e, xyz_l, dim=1)
    angle = (cos + 1).mean()
    dist = 1 / (torch.norm(xyz_e - xyz_l, p=2, dim=1).mean() + 1e-3)
    return alpha * angle + belta * dist, angle, dist


# --------------------------------------------------------------------------------------------
# a-8 / a-9 / a-10 / a-10b  optimiser side
# --------------------------------------------------------------------------------------------
def cosine_lambda(step: int, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """transformers.get_cosine_schedule_with_warmup's lr multiplier [3p] (UADA.py:109-115)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class HFAdamW(torch.optim.Optimizer):
    """transformers==4.40.1 optimization.AdamW [3p], **parity unpinned** (SURVEY.md §8a-9):
    defaults betas=(0.9,0.999), eps=1e-6, weight_decay=0, correct_bias=True; eps is added to sqrt(v)
    BEFORE the bias correction is folded into the step size (differs from torch.optim.AdamW)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                b1, b2 = group["betas"]
                st["step"] += 1
                m.mul_(b1).add_(grad, alpha=(1.0 - b1))
                v.mul_(b2).addcmul_(grad, grad, value=1.0 - b2)
                denom = v.sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    bc1 = 1.0 - b1 ** st["step"]
                    bc2 = 1.0 - b2 ** st["step"]
                    step_size = step_size * math.sqrt(bc2) / bc1
                p.addcdiv_(m, denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=(-group["lr"] * group["weight_decay"]))


def l1_clip_(grad: torch.Tensor, max_norm: float = 1e-3) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_([patch], max_norm, norm_type=1) on one tensor (UPA.py:157) [3p torch]."""
    total = grad.abs().sum()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    grad.mul_(coef)
    return total


def pgd_step(patch: torch.Tensor, grad: torch.Tensor, alpha: float) -> torch.Tensor:
    """TMA.py:171-175."""
    return (patch - alpha * grad.sign()).clamp(0, 1)


# --------------------------------------------------------------------------------------------
# CPU baseline: one reference-style inner step of the replaced ops (K1 -> [model stand-in] -> K3 -> K2 -> K4)
# --------------------------------------------------------------------------------------------
def cpu_patch_step(images_u8, patch, opt: HFAdamW, xy, theta, geometry, gout_bf16):
    """What UADA_ddp.py:190-209 executes on the host CPU around the model call (Appendix A-D9):
    K1 forward + bf16 cast, backward of a supplied upstream gradient to the patch, AdamW step, clamp."""
    opt.zero_grad()
    out = apply_random_patch_batch(images_u8, patch, xy, theta, geometry)
    out.to(torch.bfloat16).backward(gradient=gout_bf16)
    opt.step()
    patch.data = patch.data.clamp(0, 1)
    return out
