#!/usr/bin/env python3
"""bench.py — attack-steps/sec of the UADA inner loop (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one inner-loop iteration of UADA_wrapper_ddp (UADA_ddp.py:189-209):
    host RNG draws -> K1 paste/warp (HIP) -> OpenVLA-7B-shaped bf16 forward + activation backward (PyTorch-ROCm)
    -> LM head + K3 loss statistics on the labelled rows (HIP) -> K2' patch-embed backward on the tiles under the patch (MFMA) +
    patch-grad gather (HIP) -> [RCCL all-reduce, 30 KB] -> K4 AdamW + clamp (HIP, inside the step epilogue at N=1)
Workload: bs=64 PER RANK (reference semantics, UADA_ddp.py:158 -> weak scaling), 3x50x50 patch, geometry=True, maskidx=[0], synthetic
BridgeData-shaped frames resident in HBM as u8, random-init weights of the OpenVLA-7B architecture.

OUTPUT. stdout carries exactly ONE line: a COMPACT JSON record (< 4 KB, pure ASCII, strict JSON — non-finite floats become null, no
free-text notes) with the contract's keys: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config{...} / roofline{...} / roofline_k1{...} / cpu_baseline{...} / hot_path_* / loss_finite / peak_mem_GiB /
wall_s. Everything else (per-kernel tables, the standalone kernel suite, the K1 / K2 batch sweeps, rank_shapes, per-leg CPU tables, the
per-rank and per-config step blocks in full, the environment) goes to the FULL record, a JSON file (`--full-out`, default
gpurun_out/bench_full.json); its path is in the compact line (`full_record`).

What is measured:
  * `value`: W warm-up steps, then exactly K steps between barrier + synchronize, max over ranks; UN-profiled.
  * `roofline` / `roofline_k1`: a short second pass of the same steps with the library's per-dispatch timer armed (vaa_prof_*: every
    hand-written kernel launched through hipExtLaunchKernel with its own start/stop event pair = that dispatch's begin/end timestamps). The
    dominant hand-written kernel IN-STEP is the one with the most algorithmic bytes that the step actually ran (the fused LM head
    `head_stats_kernel` when the step dispatches it, else K1's `patch_apply_tiles_kernel`).
  * N=1 only: the per-rank steps of the strong-scaling configs (bs = 32 / 16 / 8 / 4), the inner steps of BASELINE configs 2 / 4 / 5
    (`config.cfg{2,4,5}_*`: single-GPU UADA with 1/CE at bs=16 geometry off; TMA's CE gradient at bs=8; UPA with resize_patch 3x100x100 at
    bs=4 — the product loops' own `inner_step`), the standalone kernel suite + sweeps (full record; `config.k2_sweep_frac` in the line) and
    `cpu_baseline` (oracle/ref_port.py on the host cores, bounded sample).
  * N>1 (one rank per GPU, RCCL): only the two timed regions — weak (`value`) and strong (the same global batch split over the ranks,
    `config.strong_*`) — with the collective's in-step cost (`config.allreduce_us_per_step`, `config.comm_frac`); the process group is
    destroyed right after them. `--regions weak|strong|both`.
`--attack uada|tma|upa` (N=1) makes the main timed region one of the other product loops (for rocprofv3 summaries of configs 2 / 4 / 5).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LINE_LIMIT = 4096  # bytes of the stdout record
INNER_LOOP = 50    # UADA_wrapper_ddp.py:104 (--innerLoop): one full-vocabulary CE evaluation per INNER_LOOP steps (UADA_ddp.py:196-221)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bs", type=int, default=64, help="per-rank batch (UADA_ddp.py:158)")
    ap.add_argument("--patch", type=str, default="3,50,50")
    ap.add_argument("--model", type=str, default="openvla-7b", choices=["openvla-7b", "tiny", "surrogate"])
    ap.add_argument("--attack", type=str, default="uada_ddp", choices=["uada_ddp", "uada", "tma", "upa"],
                    help="the product loop whose inner step the main timed region runs (default: the headline, UADA_ddp.py:189-209); the others are N=1 diagnostics")
    ap.add_argument("--geometry", type=str, default="true", choices=["true", "false"])
    ap.add_argument("--maskidx", type=str, default="", help="comma list; default: 0 (uada*), 0..6 (tma), unused by upa's reverse-direction loss")
    ap.add_argument("--resize-patch", action="store_true", help="upa: per-image patch scale s~U(0.61,1.39) (appply_random_transform.py:113-118)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-suite", action="store_true")
    ap.add_argument("--no-per-rank", action="store_true", help="skip the bs = 32 / 16 / 8 / 4 per-rank step block of the N=1 record")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE config 2 / 4 / 5 step block of the N=1 record")
    ap.add_argument("--no-val", action="store_true", help="skip the validation-pass block of the N=1 record (config.val_ms_per_batch_bs8)")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of host CPU work for cpu_baseline")
    ap.add_argument("--regions", type=str, default="both", choices=["weak", "strong", "both"],
                    help="N > 1: which timed regions run (weak = bs per rank, the headline; strong = the global batch split over the ranks). "
                         "`strong` alone is a functional / diagnostic run: `value` is then null")
    ap.add_argument("--profile-steps", type=int, default=-1, help="steps of the separate per-dispatch-profiled pass (-1: min(steps, 10); 0: none)")
    ap.add_argument("--full-out", type=str, default=os.path.join("gpurun_out", "bench_full.json"), help="where the FULL record goes (relative to the repo root)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------------------------
# the stdout record: compact, strict JSON
# ------------------------------------------------------------------------------------------------------------------------------------
def _clean(o, sig=6):
    """Strict-JSON view of a record: floats rounded to `sig` significant digits, non-finite floats -> None, numpy scalars -> python."""
    if isinstance(o, dict):
        return {str(k): _clean(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v, sig) for v in o]
    if isinstance(o, (bool, type(None), str)):
        return o
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        f = float(o)
        if not math.isfinite(f):
            return None
        if f == 0.0:
            return 0.0
        return float(f"{f:.{sig}g}")
    return str(o)


def _refuse_constant(name):
    raise ValueError(f"non-finite JSON constant {name!r}")


def encode_line(rec, optional=()):
    """The ONE stdout line: ASCII, < LINE_LIMIT bytes, no NaN / Infinity (as tokens or inside strings); keys named in `optional`
    (dotted paths, least important first) are dropped one by one if the record would not fit."""
    rec = _clean(rec)
    optional = list(optional)
    while True:
        line = json.dumps(rec, allow_nan=False, ensure_ascii=True, separators=(",", ":"))
        if len(line) < LINE_LIMIT or not optional:
            break
        path = optional.pop(0).split(".")
        d = rec
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
            if d is None:
                break
        if isinstance(d, dict):
            d.pop(path[-1], None)
    if len(line) >= LINE_LIMIT:
        raise RuntimeError(f"bench record is {len(line)} bytes (limit {LINE_LIMIT})")
    if not line.isascii() or "NaN" in line or "Infinity" in line or "\n" in line:
        raise RuntimeError("bench record is not a strict single-line ASCII JSON object")
    json.loads(line, parse_constant=_refuse_constant)
    return line


def _tunable_entries() -> int:
    """hipBLASLt/rocBLAS selections in effect (roboticattack_amd/tunableop/*.csv; 0 = the file was rejected by a validator mismatch or
    TunableOp is off -> the GEMMs run on default heuristics, ~6 % slower)."""
    try:
        import torch.cuda.tunable as tunable

        return len(tunable.get_results()) if tunable.is_enabled() else 0
    except Exception:
        return -1


def build_model(kind, dev):
    from roboticattack_amd.openvla_model import build_openvla, openvla_7b_cfg, tiny_cfg

    if kind == "openvla-7b":
        return build_openvla(openvla_7b_cfg(), device=dev, dtype=torch.bfloat16, seed=0), "OpenVLA-7B shape, random init"
    if kind == "tiny":
        return build_openvla(tiny_cfg(), device=dev, dtype=torch.bfloat16, seed=0), "tiny topology-equal model (bf16)"
    from roboticattack_amd.surrogate import SurrogateVLA

    return SurrogateVLA(seed=0).to(dev), "fp32 surrogate"


# ------------------------------------------------------------------------------------------------------------------------------------
# cpu_baseline: the reference's CPU op sequence (oracle/ref_port.py) on the host cores — the ONLY place bench.py touches oracle/
# ------------------------------------------------------------------------------------------------------------------------------------
def _cpu_leg(threads, bs, patch_shape, budget_s, conn):
    """One leg of the CPU baseline in its own process (so that a thread count that thrashes can be stopped by the parent)."""
    import numpy as _np
    import torch as _torch

    from oracle import ref_port
    from roboticattack_amd import synthetic
    from roboticattack_amd.benchmarks import random_params

    _torch.set_num_threads(threads)
    n_max = min(8, bs) if threads == 1 else bs
    gout_all = synthetic.synth_upstream_grad(7, n_max)
    imgs_all = synthetic.synth_images(1234, n_max, "noise")
    xy_all, th_all = random_params(n_max, patch_shape[1], patch_shape[2], 42)
    th_all = th_all.reshape(n_max, 2, 3)
    _, labels_all, _ = synthetic.synth_text_batch(4242, bs)
    labels_all = ref_port.mask_labels(labels_all, [0])[:n_max]
    S = 256 + labels_all.shape[1]
    logits4 = _torch.randn(4, S, 32064)

    def patch_ops(n):
        patch = _torch.nn.Parameter(_torch.rand(*patch_shape))
        opt = ref_port.HFAdamW([patch], lr=1e-3)
        t0 = time.perf_counter()
        ref_port.cpu_patch_step(imgs_all[:n], patch, opt, xy_all[:n], th_all[:n], True, gout_all[:n])
        return time.perf_counter() - t0

    def loss_ops(n):
        lg = logits4.repeat((n + 3) // 4, 1, 1)[:n].contiguous().requires_grad_(True)  # values do not affect timing
        t0 = time.perf_counter()
        mse, _ = ref_port.uada_weighted_loss(lg, labels_all[:n], 5.0)
        ce = ref_port.hf_ce(lg, labels_all[:n])  # HF computes `.loss` whenever labels are passed (UADA_ddp.py:196-201)
        (mse + 0.0 * ce).backward()
        return time.perf_counter() - t0

    def timed(fn, t_budget, warm=2, iters=5):
        """2 warm-ups + 5 timed iterations that fit the budget: a 2-image probe sizes the sample (every op of the path is linear in the batch)."""
        n0 = min(2, n_max)
        fn(n0)
        per_img = fn(n0) / n0
        n = max(1, min(n_max, int(0.6 * t_budget / (warm + iters) / max(per_img, 1e-9))))
        for _ in range(warm):
            fn(n)
        ts = [fn(n) for _ in range(iters)]
        return float(_np.min(ts)), float(_np.median(ts)), n

    p_min, p_med, p_n = timed(patch_ops, budget_s * 0.2)
    l_min, l_med, l_n = timed(loss_ops, budget_s * 0.8)
    conn.send({"threads": threads, "sample_bs_K1_K2_K4": p_n, "sample_bs_K3": l_n, "scaled_to_bs": bs, "logits_S": S,
               "ms_K1_K2_K4_min": p_min * bs / p_n * 1e3, "ms_K1_K2_K4_median": p_med * bs / p_n * 1e3,
               "ms_K3_min": l_min * bs / l_n * 1e3, "ms_K3_median": l_med * bs / l_n * 1e3,
               "steps_per_s_min_time": 1.0 / (p_min * bs / p_n + l_min * bs / l_n),
               "steps_per_s_median_time": 1.0 / (p_med * bs / p_n + l_med * bs / l_n)})
    conn.close()


def cpu_baseline(bs, patch_shape, budget_s=25.0):
    """The reference's CPU path for the replaced ops (oracle/ref_port.py: per-image PyTorch op chain + autograd for K1/K2, HF-style CE +
    weighted_loss on fp32 logits [B,S,32064] + backward for K3, HF AdamW + clamp for K4), timed on this box's host cores with ONE thread
    and with ALL logical cores: 2 warm-ups, then 5 timed iterations, min and median.

    Bounded: every leg times a slice of the per-rank batch whose size a 2-image probe picks so that 2 + 5 iterations fit the leg's
    budget and scales linearly to bs; every leg runs in its own process under a wall-clock guard, because torch.set_num_threads(256)
    thrashes on the reference's many tiny per-image ops: a leg that exceeds its guard is reported as timed out. Hosts with more than 32
    logical cores also get a 32-thread leg (the setting the CPU path ran best at in round 1).
    Returns (compact object for the stdout line, full object with every leg)."""
    import multiprocessing as mp

    ncores = os.cpu_count() or 1
    ctx = mp.get_context("spawn")
    plan = [("1_thread", 1), ("all_cores", ncores)] + ([("32_threads", 32)] if ncores > 32 else [])
    leg_budget = budget_s / len(plan)
    legs = {}
    for tag, threads in plan:
        rx, tx = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_cpu_leg, args=(threads, bs, patch_shape, leg_budget, tx))
        p.start()
        tx.close()
        guard = 4.0 * leg_budget + 30.0  # import + probe + iterations; generous, but finite
        try:
            if rx.poll(guard):
                legs[tag] = rx.recv()
                p.join(10)
            else:
                legs[tag] = {"threads": threads, "timed_out_after_s": guard}
        except EOFError:
            legs[tag] = {"threads": threads, "failed": "the leg's process exited without a result"}
        if p.is_alive():
            p.terminate()
            p.join(5)
    ok = {k: v for k, v in legs.items() if "steps_per_s_min_time" in v}
    best = max(ok, key=lambda k: ok[k]["steps_per_s_min_time"]) if ok else None
    bl = ok[best] if best else {}
    compact = {"value": bl.get("steps_per_s_min_time"), "unit": "patch-path steps/s (K1+K2+K3+K4 on host CPU, model excluded)",
               "cores": bl.get("threads", 0), "kind": "port",
               "sample": f"oracle/ref_port.py, slice of the bs={bs} batch sized by a 2-image probe (K1K2K4 {bl.get('sample_bs_K1_K2_K4')} img, "
                         f"K3 {bl.get('sample_bs_K3')} img), scaled linearly to bs={bs}; best of {len(plan)} thread counts, min of 5",
               "host_logical_cores": ncores, "ms_K1_K2_K4": bl.get("ms_K1_K2_K4_min"), "ms_K3": bl.get("ms_K3_min")}
    full = dict(compact, legs=legs)
    return compact, full


# ------------------------------------------------------------------------------------------------------------------------------------
# step runners: the product loops' own inner steps on synthetic frames resident in HBM
# ------------------------------------------------------------------------------------------------------------------------------------
class StepRunner:
    """One rank's attack-loop state for a per-rank batch of B images: the inner step of attack/uada_ddp.py — the SAME code
    (AttackBase.fused_ddp_step / model_loss) — on synthetic frames resident in HBM."""

    name = "uada_ddp"

    def __init__(self, model, dev, B, patch_shape, rank, world, geometry=True, maskidx=(0,)):
        from roboticattack_amd import dist as vdist
        from roboticattack_amd import ops, synthetic
        from roboticattack_amd.attack.engine import AttackBase
        from roboticattack_amd.labels import mask_labels
        from roboticattack_amd.optim import PatchOptimizer

        self.ops, self.model, self.dev, self.B, self.world, self.geometry = ops, model, dev, B, world, bool(geometry)
        self.att = AttackBase(model, None, "", "adamW", False)  # K2'  when the model exposes its patch-embed weights; VAA_FUSED_EMBED_GRAD=0: plain K2
        self.use_rows = self.att.use_rows
        self.tr = self.att.randomPatchTransform
        self.fused = self.att.fused_ddp_available()  # K2's final sum + K3's fold + the DDP message in one launch (VAA_FUSED_EPILOGUE=0: separate launches)
        self.batch = synthetic.synth_batch(1234 + rank, B, "noise", as_pil=False)
        self.img = self.tr.stage_images(torch.from_numpy(self.batch["pixel_values"]))
        self.input_ids = self.batch["input_ids"].to(dev)
        self.attn = self.batch["attention_mask"].to(dev)
        self.labels = mask_labels(self.batch["labels"].clone(), list(maskidx)).to(dev)
        torch.manual_seed(42)  # UADA_wrapper_ddp.py:53: every rank seeds 42
        patch = torch.rand(patch_shape).to(dev) if rank == 0 else torch.empty(patch_shape).to(dev)
        vdist.broadcast_patch(patch)
        self.patch = patch.requires_grad_(True)
        self.opt = PatchOptimizer(self.patch, 1e-3, "adamW")
        self.sync = vdist.PatchGradSync(self.patch.numel(), 4, dev)
        self.inv_world = 1.0 / world
        self.scal = torch.zeros(8, device=dev)
        self.pick = torch.tensor([1, 2, 7, 0], dtype=torch.int64, device=dev)
        self.R = int((self.labels[:, 1:] != -100).sum())
        # the loop's cadence (attack/uada_ddp.py = UADA_ddp.py:196-221): the full-vocabulary CE / argmax is read on the LAST inner step of an
        # outer iteration only, innerLoop = 50 (UADA_wrapper_ddp.py:104): steps count through outer iterations, warm-up included
        self.full_ce_every = INNER_LOOP
        self.k = 0
        self.force_ce = None  # the per-dispatch-profiled blocks pin the step kind (True / False); None = the cadence

    def step(self):  # attack/uada_ddp.py inner step
        a, ops = self.att, self.ops
        full_ce = ((self.k + 1) % self.full_ce_every == 0) if self.force_ce is None else bool(self.force_ce)
        self.k += 1
        self.ce_steps = getattr(self, "ce_steps", 0) + int(full_ce)
        self.opt.zero_grad()
        if self.fused and self.world == 1:
            # host draws -> K1 (tile-major) -> model -> K3s (+ K3h on CE steps) -> backward -> K2' tiles + scatter -> epilogue incl. K4 (nothing to exchange)
            a.fused_ddp_step(self.img, self.patch, self.input_ids, self.attn, self.labels, self.geometry, 5.0, self.sync.buf, self.scal, optimizer=self.opt,
                             full_ce=full_ce)
            return
        if self.fused:
            # ... -> epilogue: the message is in sync.buf -> all-reduce -> K4
            a.fused_ddp_step(self.img, self.patch, self.input_ids, self.attn, self.labels, self.geometry, 5.0, self.sync.buf, self.scal, full_ce=full_ce)
            g_sum, _ = self.sync.allreduce_packed()  # [grad | CE, MSE, UAD, total]: one all-reduce per step
        else:
            pix = self.tr.apply_random_patch_batch(self.img, self.patch, mean=a.mean, std=a.std, geometry=self.geometry)  # host RNG draws + K1
            total, scalars, _ = a.model_loss(self.input_ids, self.attn, pix, self.labels, ops.LOSS_UADA_DDP, w=5.0, full_ce=full_ce)  # model + LM head + K3
            total.backward()  # ... -> K2 (or K2' fed by the patch-embed output gradients)
            g_sum, _ = self.sync.allreduce_step(self.patch.grad, scalars, self.pick)
            self.scal.copy_(scalars)
        self.opt.step(grad=g_sum.view_as(self.patch), grad_scale=self.inv_world)  # K4


class LoopRunner:
    """The inner step of one of the single-GPU product loops — attack/uada.py (UADA.py:133-159), attack/tma.py (TMA.py:124-175),
    attack/upa.py (UPA.py:127-159) — called through the loop's own `inner_step`, on synthetic frames resident in HBM (N=1)."""

    def __init__(self, kind, model, dev, B, patch_shape, geometry=True, maskidx=None, resize_patch=False):
        from roboticattack_amd import ops, synthetic
        from roboticattack_amd.labels import mask_labels, tma_target_labels, tma_target_tokens
        from roboticattack_amd.optim import PatchOptimizer

        self.name, self.ops, self.model, self.dev, self.B, self.world, self.geometry = kind, ops, model, dev, B, 1, bool(geometry)
        batch = synthetic.synth_batch(1234, B, "noise", as_pil=False)
        self.batch = batch
        labels = batch["labels"].clone()
        l1 = 0.0
        if kind == "uada":
            from roboticattack_amd.attack.uada import OpenVLAAttacker

            self.att = OpenVLAAttacker(model, None, "", "adamW", False)
            self.maskidx = list(maskidx) if maskidx is not None else [0]
            labels = mask_labels(labels, self.maskidx)
            lr = 1e-3  # UADA_wrapper.py:93
        elif kind == "tma":
            from roboticattack_amd.attack.tma import OpenVLAAttacker

            self.att = OpenVLAAttacker(model, None, "", "adamW", False)
            self.maskidx = list(maskidx) if maskidx is not None else list(range(7))  # BASELINE config 4: a 7-DoF target vector
            target = tma_target_tokens(np.zeros(7), self.maskidx, self.att.action_tokenizer)  # targetAction 0 (TMA_wrapper.py:123)
            labels = tma_target_labels(labels, target)
            lr = 2e-3  # TMA_wrapper.py:95
        elif kind == "upa":
            from roboticattack_amd.attack.upa import OpenVLAAttacker

            self.att = OpenVLAAttacker(model, None, "", "adamW", bool(resize_patch), 0.8, 0.2)
            self.maskidx = list(maskidx) if maskidx is not None else [0, 1, 2]
            self.mode, self.scale = self.att._mode(False, True)  # reverse_direction=True (UPA_wrapper.py default): labels stay unmasked
            lr, l1 = 2e-3, 1e-3  # UPA_wrapper.py:96, UPA.py:157
        else:
            raise ValueError(kind)
        self.use_rows = self.att.use_rows
        self.tr = self.att.randomPatchTransform
        self.img = self.tr.stage_images(torch.from_numpy(batch["pixel_values"]))
        self.input_ids, self.attn, self.labels = batch["input_ids"].to(dev), batch["attention_mask"].to(dev), labels.to(dev)
        torch.manual_seed(42)
        self.patch = torch.rand(patch_shape).to(dev).requires_grad_(True)
        self.opt = PatchOptimizer(self.patch, lr, "adamW", l1_clip=l1)
        self.scal10 = torch.zeros((1, 10), dtype=torch.float32, device=dev)
        self.R = int((self.labels[:, 1:] != -100).sum())
        self.sync = None
        self.k = 0
        self.read_every = 100  # UPA_wrapper.py:119 (--innerLoop): the loop reads the loss terms of the LAST inner step of an outer iteration (UPA.py:171-186)
        self.force_ce = None   # True / False pins the step kind (profiled blocks, the finite-state check); None = the cadence

    @property
    def scal(self):
        return self.scal10[0, :8]

    def step(self):
        a = self.att
        self.k += 1
        if self.name == "uada":
            a.inner_step(self.patch, self.opt, self.img, self.input_ids, self.attn, self.labels, self.geometry, self.scal10, 0)
        elif self.name == "tma":
            a.inner_step(self.patch, self.opt, self.img, self.input_ids, self.attn, self.labels, self.geometry, False, self.scal10, 0)
        else:
            read = (self.k % self.read_every == 0) if self.force_ce is None else bool(self.force_ce)
            a.inner_step(self.patch, self.opt, self.img, self.input_ids, self.attn, self.labels, self.geometry, self.mode, self.scale, self.scal10, 0, read_scalars=read)


def make_runner(kind, model, dev, B, patch_shape, rank, world, geometry=True, maskidx=None, resize_patch=False):
    if kind == "uada_ddp":
        return StepRunner(model, dev, B, patch_shape, rank, world, geometry=geometry, maskidx=maskidx if maskidx is not None else (0,))
    if world != 1:
        raise SystemExit("--attack uada / tma / upa are single-GPU loops of the reference (N=1 diagnostics); N > 1 runs the data-parallel UADA step")
    return LoopRunner(kind, model, dev, B, patch_shape, geometry=geometry, maskidx=maskidx, resize_patch=resize_patch)


def comm_summary(recs, world, steps, dt, dev):
    """In-step cost of the gradient exchange from the per-call event brackets of PatchGradSync (one all-reduce per step; events on the launch
    stream around dist.all_reduce of the [grad | 4 scalars] message, 30,016 B at 50x50): rank 0's own figures plus min / max over ranks of
    the per-rank means — the rank that reaches the collective LAST sees the exchange alone, the others also wait for it (skew), so
    min ~ hand-off + exchange latency (comm_frac = min / step time) and max - min ~ rank skew (comm_plus_skew_frac = max / step time)."""
    import torch.distributed as dist

    if world == 1 or not recs:
        return None
    ev = np.asarray([r[0] for r in recs], dtype=np.float64)
    host = np.asarray([r[1] for r in recs], dtype=np.float64)
    t = torch.tensor([ev.mean(), -ev.mean(), ev.max()], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    mean_max, mean_min, single_max = float(t[0]), -float(t[1]), float(t[2])
    step_us = dt / steps * 1e6
    return {"calls_per_step": len(recs) / steps, "mean_us_rank0": float(ev.mean()), "median_us_rank0": float(np.median(ev)), "max_us_rank0": float(ev.max()),
            "host_us_rank0": float(host.mean()), "mean_us_min_over_ranks": mean_min, "mean_us_max_over_ranks": mean_max, "max_us_any_rank": single_max,
            "comm_frac": mean_min / step_us, "comm_plus_skew_frac": mean_max / step_us}


def allreduce_back_to_back(sync, world, n=50):
    """The collective alone: n all-reduces of the step's message back to back between two events (no compute in between, ranks aligned by
    the collectives themselves) -> mean us per call."""
    import torch.distributed as dist

    if world == 1:
        return None
    for _ in range(5):
        dist.all_reduce(sync.buf, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        dist.all_reduce(sync.buf, op=dist.ReduceOp.SUM)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    sync.buf.zero_()  # n sums of a non-zero message can overflow; the next step rewrites the buffer anyway
    return {"calls": n, "event_us_per_call": e0.elapsed_time(e1) * 1e3 / n, "wall_us_per_call": wall, "bytes": int(sync.buf.numel() * 4)}


def timed_steps(runner, steps, warmup, world, dev, profile=False, comm=False):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; returns the max over ranks (s), host costs, the
    per-dispatch records (profile=True) and the per-call all-reduce brackets (comm=True)."""
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        runner.step()
    barrier()
    if profile:
        runner.ops.prof_start(64 * steps)  # per-dispatch start/stop events on every hand-written kernel of the timed region
    if comm and world > 1:
        runner.sync.timing_start()
    t0 = time.perf_counter()
    host_enqueue, cpu0 = 0.0, time.thread_time()
    for _ in range(steps):
        th0 = time.perf_counter()
        runner.step()
        host_enqueue += time.perf_counter() - th0  # wall time the host spends inside step() (no explicit sync inside a step)
    host_cpu = time.thread_time() - cpu0  # CPU time of the launching thread: the real host cost (enqueue wall time also contains back-pressure waits)
    barrier()
    dt = time.perf_counter() - t0
    recs = runner.ops.prof_collect() if profile else []
    crecs = runner.sync.timing_collect() if (comm and world > 1) else []
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), host_enqueue / steps, host_cpu / steps, recs, crecs


# kernel name (substring of the launch site's name) -> operator of the hot path
KERNEL_OPS = (("patch_apply_fwd_kernel", "K1"), ("patch_apply_tiles_kernel", "K1"), ("patch_apply", "K1"), ("embed_dgrad", "K2e"), ("patch_grad_scatter", "K2"), ("patch_grad_reduce", "K2"),
              ("patch_grad", "K2"), ("head_slice", "K3s"), ("head_stats_kernel", "K3h"), ("head_finish_kernel", "K3h"), ("rows_stats_kernel", "K3"), ("rows_finish_kernel", "K3"), ("rows_", "K3"),
              ("loss_stats_kernel", "K3"), ("loss_grad_kernel", "K3"), ("loss_", "K3"), ("step_epilogue_kernel", "EPI"), ("patch_update_kernel", "K4"), ("patch_resize", "K0"), ("resize", "K0"))


def op_of(name):
    return next((o for sub, o in KERNEL_OPS if sub in name), "other")


def kernel_table(recs, psteps):
    """per-kernel durations of a per-dispatch-profiled pass: every dispatch's own begin/end timestamps (vaa_prof_*)"""
    per = {}
    for name, us in recs:
        per.setdefault(name, []).append(us)
    kern, op_us = {}, {}
    for name, ts in per.items():
        t = np.asarray(ts)
        kern[name] = {"op": op_of(name), "launches": len(ts), "launches_per_step": len(ts) / psteps, "mean_us": float(t.mean()), "median_us": float(np.median(t)),
                      "min_us": float(t.min()), "max_us": float(t.max()), "us_per_step": float(t.sum()) / psteps}
        op_us[kern[name]["op"]] = op_us.get(kern[name]["op"], 0.0) + float(t.sum()) / psteps
    return kern, op_us


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the driver's N > 1 form passes
    the rendezvous itself and never gets here). With fewer visible GPUs than ranks the run falls back to gloo with ranks sharing GPUs —
    a functional run (tests), flagged in the record, not a scaling measurement."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    if torch.cuda.device_count() < args.gpus and not env.get("VAA_DIST_BACKEND"):
        print(f"bench.py: {torch.cuda.device_count()} GPU(s) visible for --gpus {args.gpus}: ranks share GPUs over gloo (functional run only)", file=sys.stderr)
        env["VAA_DIST_BACKEND"] = "gloo"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


# ------------------------------------------------------------------------------------------------------------------------------------
# N=1 blocks beside the headline
# ------------------------------------------------------------------------------------------------------------------------------------
def head_bytes(cfg):
    return 2 * 32064 * cfg.llm_dim if cfg is not None and hasattr(cfg, "llm_dim") else None


def per_rank_block(model, dev, B, patch_shape, steps, ips_ref):
    """The per-rank step of the strong-scaling configs on ONE GPU: what each of 2 / 4 / 8 ranks runs in config 3 (bs = 32 / 16 / 8; config 4:
    32 over 4 -> 8) and config 5 (32 over 8 -> 4). projected speedup = ranks x images/s at that batch / images/s at bs=B, i.e. the
    strong-scaling ceiling before the 30 KB all-reduce."""
    from roboticattack_amd.benchmarks import HBM_PEAK_GBS

    out = {}
    hb = head_bytes(getattr(model, "cfg", None))
    for b in (32, 16, 8, 4):
        if b >= B:
            continue
        torch.cuda.empty_cache()
        rb = StepRunner(model, dev, b, patch_shape, 0, 1)
        n_b = 6  # (the blocks beside the headline are 6 timed steps each)
        dt_b, enq_b, cpu_b, _, _ = timed_steps(rb, n_b, 3, 1, dev)  # un-profiled, like the headline it is compared with
        n_p = min(n_b, 5)
        _, _, _, recs_b, _ = timed_steps(rb, n_p, 0, 1, dev, profile=True)
        kern_b, hot_b = kernel_table(recs_b, n_p)
        e = {"ms_per_step": dt_b / n_b * 1e3, "images_per_s": b * n_b / dt_b, "images_per_s_vs_bs%d" % B: (b * n_b / dt_b) / ips_ref,
             "projected_speedup_%d_ranks_before_comm" % (B // b): (B // b) * (b * n_b / dt_b) / ips_ref,
             "host_cpu_ms_per_step": cpu_b * 1e3, "host_enqueue_ms_per_step": enq_b * 1e3, "labelled_rows": rb.R,
             "hot_path_us_per_step": sum(hot_b.values()), "hot_path_ops_us": hot_b}
        hk = next((k for n, k in kern_b.items() if "head_stats_kernel" in n), None)
        if hk and hb:  # the fused LM head (a 263 MB weight stream at the 7B shape): in-step duration and HBM rate
            e["fused_head"] = {"kernel": "head_stats_kernel", "mean_us": hk["mean_us"], "algo_bytes": hb, "achieved_GBs": hb / hk["mean_us"] / 1e3,
                               "frac": hb / hk["mean_us"] / 1e3 / HBM_PEAK_GBS}
        out[f"bs{b}"] = e
        del rb
    return out


def val_block(model, dev, bs=8, batches=12):
    """The validation pass of the data-parallel loop (attack/uada_ddp.py validate = UADA_ddp.py:233-281: no-grad forwards over the rank's validation
    batches, bs = 8 per rank in BASELINE config 3, K1 -> model -> K3h with the CE that is logged) through the PRODUCT's own `validate`, un-synchronised
    (one read-back behind the last batch) and with the round-5 per-batch read-back (VAA_VAL_SYNC_EVERY_BATCH=1): ms per batch, host included."""
    from roboticattack_amd.attack import uada_ddp
    from roboticattack_amd.synthetic import SyntheticLoader

    class Att(uada_ddp.OpenVLAAttacker):
        val_batches = batches

    att = Att(vla_path="resident", dataset_name="synthetic", save_dir="", patch_size=[3, 50, 50], lr=1e-3, bs=bs, warmup=20, num_iter=1, maskidx=[0], innerLoop=1,
              geometry=True, use_wandb=False, MSE_weights=5, device=dev, model_factory=lambda path, d: model,
              dataset_factory=lambda name, b, rank, world: (SyntheticLoader(b, seed=1, kind="noise"), SyntheticLoader(b, seed=2, kind="noise")))
    patch = torch.rand(3, 50, 50, device=dev)
    out = {}
    for tag, env in (("sync_every_batch", "1"), ("one_readback", "0"), ("sync_every_batch_2", "1"), ("one_readback_2", "0")):
        os.environ["VAA_VAL_SYNC_EVERY_BATCH"] = env
        att.validate(0, patch, rank=1)  # warm (rank != 0: nothing is written)
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.thread_time()
        att.validate(0, patch, rank=1)
        torch.cuda.synchronize()
        out[tag] = {"ms_per_batch": (time.perf_counter() - t0) / batches * 1e3, "host_cpu_ms_per_batch": (time.thread_time() - c0) / batches * 1e3}
    os.environ.pop("VAA_VAL_SYNC_EVERY_BATCH", None)
    return {"bs": bs, "batches": batches, "what": "attack/uada_ddp.py validate (UADA_ddp.py:233-281), A B A B", **out,
            "ms_per_batch": min(out["one_readback"]["ms_per_batch"], out["one_readback_2"]["ms_per_batch"]),
            "ms_per_batch_sync_every_batch": min(out["sync_every_batch"]["ms_per_batch"], out["sync_every_batch_2"]["ms_per_batch"])}


CONFIG_STEPS = (
    # tag, loop, per-rank batch, patch, geometry, resize_patch, what BASELINE.json calls it
    ("cfg2", "uada", 16, [3, 50, 50], False, False, "UADA single-GPU: bs=16, geometry=False, loss = MSE + 1/CE (UADA.py:133-159)"),
    ("cfg4", "tma", 8, [3, 50, 50], True, False, "TMA per-rank step: bs=8 (32 over 4 GPUs), 7-DoF target, CE gradient (TMA.py:124-175)"),
    ("cfg5", "upa", 4, [3, 100, 100], True, True, "UPA per-rank step: bs=4 (32 over 8 GPUs), resize_patch 3x100x100 (UPA.py:127-159)"),
)


def config_block(model, dev, steps):
    """The inner steps of BASELINE configs 2 / 4 / 5 at their per-rank shapes through the product loops' own `inner_step`: ms per step
    (un-profiled), then the hand-written launches of the step per dispatch (per kernel and per operator) and the dominant hand-written
    kernel (most algorithmic bytes) with its HBM fraction."""
    out = {}
    for tag, kind, b, pshape, geo, resize, what in CONFIG_STEPS:
        torch.cuda.empty_cache()
        r = LoopRunner(kind, model, dev, b, pshape, geometry=geo, resize_patch=resize)
        n = 6
        dt, enq, cpu, _, _ = timed_steps(r, n, 3, 1, dev)
        n_p = 5
        _, _, _, recs, _ = timed_steps(r, n_p, 0, 1, dev, profile=True)
        kern, op_us = kernel_table(recs, n_p)
        r.force_ce = True
        r.step()  # a step that reads its scalars (UPA folds them on the last inner step of an outer iteration only)
        r.force_ce = None
        e = {"what": what, "loop": kind, "bs": b, "patch": pshape, "geometry": geo, "resize_patch": resize, "labelled_rows": r.R,
             "ms_per_step": dt / n * 1e3, "steps_per_s": n / dt, "images_per_s": b * n / dt, "host_cpu_ms_per_step": cpu * 1e3,
             "hot_path_us_per_step": sum(op_us.values()), "hot_path_launches_per_step": sum(k["launches_per_step"] for k in kern.values()),
             "hot_path_ops_us": op_us, "kernels": kern, "loss_finite": bool(torch.isfinite(r.scal).all())}
        e["dominant"] = dominant_kernel(kern, model, r, b, pshape)
        out[tag] = e
        del r
    return out


def kernel_bytes(name, model, runner, B, pshape):
    """Algorithmic bytes per launch of a hand-written kernel of the step, by its launch-site name (None: a latency-bound helper)."""
    from roboticattack_amd.benchmarks import algo_bytes

    cfg = getattr(model, "cfg", None)
    R = runner.R
    ph, pw = pshape[1], pshape[2]
    if getattr(runner, "tr", None) is not None and runner.tr.resize_patch and runner.tr.last_sizes is not None:
        ph, pw = [int(v) for v in np.asarray(runner.tr.last_sizes).mean(axis=0)]  # per-image sizes: the mean footprint of the last draw
    width = (cfg.dino.dim + cfg.siglip.dim) if cfg is not None and hasattr(cfg, "dino") else 1024 + 1152
    if "patch_apply" in name:
        return algo_bytes("K1", B, ph, pw)
    if "head_slice_kernel" in name and cfg is not None and hasattr(cfg, "llm_dim"):
        return 2 * (2 * R * cfg.llm_dim) + 2 * (2 * 256 * cfg.llm_dim)  # hidden rows in, d hidden out, the weight slice and its transposed copy
    if "head_stats_kernel" in name and head_bytes(cfg):
        return head_bytes(cfg) + 2 * R * cfg.llm_dim
    if "embed_dgrad" in name or ("patch_grad_scatter" in name and getattr(runner.tr, "embed_with", None) is not None):
        return algo_bytes("K2e", B, ph, pw, embed_width=width)  # the op's bytes: quoted on the tile GEMM, its scatter moves the tile gradients again
    if "patch_grad" in name:
        return algo_bytes("K2", B, ph, pw)
    if "rows_stats_kernel" in name or "rows_finish_kernel" in name or "loss_stats_kernel" in name or "loss_grad_kernel" in name:
        esz = 2 if cfg is not None else 4
        if runner.name in ("uada", "tma"):  # CE gradient over every logit of the labelled rows: the statistics pass reads them, the finish pass reads + writes
            return algo_bytes("K3", B, rows=R, esize=esz) * (0.5 if "stats" in name else 1.0)
        return algo_bytes("K3_slice", B, rows=R, esize=esz) if "stats" in name else None
    return None


def dominant_kernel(kern, model, runner, B, pshape):
    from roboticattack_amd.benchmarks import HBM_PEAK_GBS

    best = None
    for name, k in kern.items():
        nb = kernel_bytes(name, model, runner, B, pshape)
        if nb and (best is None or nb > best["algo_bytes"]):
            best = {"kernel": name.strip("() "), "op": k["op"], "mean_us": k["mean_us"], "algo_bytes": int(nb), "achieved_GBs": nb / k["mean_us"] / 1e3,
                    "frac": nb / k["mean_us"] / 1e3 / HBM_PEAK_GBS, "launches_per_step": k["launches_per_step"]}
    return best


def write_full(path, full):
    """the FULL record -> a JSON file (strict JSON too); returns the repo-relative path or None when it cannot be written"""
    p = path if os.path.isabs(path) else os.path.join(ROOT, path)
    try:
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            json.dump(_clean(full, sig=8), f, allow_nan=False, indent=1)
            f.write("\n")
        print(f"bench.py: full record -> {p}", file=sys.stderr)
        return os.path.relpath(p, ROOT)
    except OSError as e:
        print(f"bench.py: could not write the full record to {p}: {e}", file=sys.stderr)
        return None


def main():
    t_main = time.perf_counter()
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    # stdout carries ONE line, the JSON record: libraries that print banners to the process's stdout (RCCL / gloo at init, MIOpen, ...)
    # are sent to stderr for the duration of the run; the original stdout is restored for the record
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist

    from roboticattack_amd import dist as vdist
    from roboticattack_amd import ops
    from roboticattack_amd.benchmarks import HBM_PEAK_GBS, k2_sweep, kernel_suite

    ops.device_check()
    dev = vdist.local_device()
    torch.cuda.set_device(dev)
    if world > 1:
        vdist.init_process_group(device=dev)  # "nccl" (= RCCL) on GPUs; VAA_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
    random.seed(42)
    np.random.seed(42)
    torch.manual_seed(42)  # UADA_wrapper_ddp.py:53: every rank seeds 42

    patch_shape = [int(v) for v in args.patch.split(",")]
    B = args.bs
    geometry = args.geometry == "true"
    maskidx = [int(v) for v in args.maskidx.split(",")] if args.maskidx else None
    model, model_desc = build_model(args.model, dev)
    runner = make_runner(args.attack, model, dev, B, patch_shape, rank, world, geometry=geometry, maskidx=maskidx, resize_patch=args.resize_patch)
    use_rows, tr, R = runner.use_rows, runner.tr, runner.R
    headline = args.attack == "uada_ddp"

    # PCIe leg, outside the timed region: staging the batch's frames (host list of uint8 HWC arrays -> HBM), done once per
    # OUTER iteration by the attack loops; the reference re-does ToTensor + H2D for every image on every inner step.
    host_frames = [np.ascontiguousarray(f) for f in runner.batch["pixel_values"]]
    tr.stage_images(host_frames)
    torch.cuda.synchronize()
    t_stage = time.perf_counter()
    tr._staged_key = None
    tr.stage_images(host_frames)
    torch.cuda.synchronize()
    stage_ms = (time.perf_counter() - t_stage) * 1e3

    run_weak = world == 1 or args.regions in ("weak", "both")
    run_strong = world > 1 and args.regions in ("strong", "both")
    psteps = min(args.steps, 10) if args.profile_steps < 0 else args.profile_steps

    # ---- the timed region: weak scaling, bs = args.bs PER RANK (reference semantics, UADA_ddp.py:158). UN-profiled, like the strong region ----
    t_setup = time.perf_counter() - t_main
    dt = host_enqueue = host_cpu = None
    recs, recs_ce, comm_w, finite = [], [], None, True
    if run_weak:
        dt, host_enqueue, host_cpu, _, crecs = timed_steps(runner, args.steps, args.warmup, world, dev, comm=True)
        comm_w = comm_summary(crecs, world, args.steps, dt, dev)
        # ---- the same steps once more, SEPARATELY, with the library's per-dispatch timer armed: the in-step kernel durations, per step KIND
        #      (slice-only steps, then steps that also evaluate the full-vocabulary CE: 1 of INNER_LOOP in the loop) ----
        if psteps > 0:
            cadenced = hasattr(runner, "force_ce")
            if cadenced:
                runner.force_ce = False
            _, _, _, recs, _ = timed_steps(runner, psteps, 1, world, dev, profile=True)
            if cadenced:
                runner.force_ce = True
                _, _, _, recs_ce, _ = timed_steps(runner, min(psteps, 4), 1, world, dev, profile=True)
                runner.force_ce = None
        # the loss scalars of a step that READS them (the loop folds them on the last inner step of an outer iteration only): finite?
        if hasattr(runner, "force_ce"):
            runner.force_ce = True
        runner.step()
        if hasattr(runner, "force_ce"):
            runner.force_ce = None
        torch.cuda.synchronize()
        finite = bool(torch.isfinite(runner.scal).all()) and bool(torch.isfinite(runner.patch.detach()).all())
    t_region = time.perf_counter() - t_main - t_setup
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2**30
    b2b = allreduce_back_to_back(runner.sync, world) if world > 1 else None

    # ---- strong scaling (BASELINE config 3: "bs=64 over N GPUs"): the same global batch split over the ranks ----
    strong = None
    if run_strong:
        bs_s = max(1, B // world)
        torch.cuda.empty_cache()  # the weak region's cached activation blocks are not needed any more (ranks that share a GPU in test mode are near its capacity)
        r_s = StepRunner(model, dev, bs_s, patch_shape, rank, world, geometry=geometry)
        dt_s, enq_s, cpu_s, _, crecs_s = timed_steps(r_s, args.steps, args.warmup, world, dev, comm=True)
        comm_s = comm_summary(crecs_s, world, args.steps, dt_s, dev)
        r_s.force_ce = True
        r_s.step()
        r_s.force_ce = None
        fin_s = torch.tensor([1.0 if bool(torch.isfinite(r_s.scal).all()) else 0.0], device=dev)
        dist.all_reduce(fin_s, op=dist.ReduceOp.MIN)
        mem_s = torch.tensor([torch.cuda.max_memory_allocated(dev) / 2**30], dtype=torch.float64, device=dev)
        dist.all_reduce(mem_s, op=dist.ReduceOp.MAX)
        strong = {"per_rank_bs": bs_s, "global_batch": bs_s * world, "ms_per_step": dt_s / args.steps * 1e3, "steps_per_s": args.steps / dt_s,
                  "images_per_s": bs_s * world * args.steps / dt_s, "host_cpu_ms_per_step": cpu_s * 1e3, "host_enqueue_ms_per_step": enq_s * 1e3,
                  "loss_finite_all_ranks": bool(fin_s.item() > 0.5), "peak_mem_GiB_max_over_ranks": float(mem_s.item()),
                  "allreduce_us_per_step": comm_s, "comm_frac": comm_s["comm_frac"] if comm_s else None}
        if run_weak:  # images/s of the split global batch over ALL ranks / images/s of ONE rank's bs=B step in this same run
            strong["speedup_vs_one_rank_weak_step"] = (bs_s * world * args.steps / dt_s) / (B * args.steps / dt)
        del r_s
    # every rank's hipBLASLt / rocBLAS selections (a rank that lost them runs ~6 % slower and drags the synchronous step): min over ranks
    tun = _tunable_entries()
    tun_min = tun
    if world > 1:
        tt = torch.tensor([float(tun)], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        tun_min = int(tt.item())
        # N > 1 is the two timed regions and nothing else: the collectives are over, every rank leaves the group now (UADA_ddp.py:214-221)
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # ---- N=1 only: the per-rank steps of the strong-scaling configs and the inner steps of BASELINE configs 2 / 4 / 5 ----
    per_rank = cfg_steps = val_rec = None
    if world == 1 and headline:
        if not args.no_val:
            torch.cuda.empty_cache()
            val_rec = val_block(model, dev)
        if not args.no_per_rank:
            per_rank = per_rank_block(model, dev, B, patch_shape, args.steps, B * args.steps / dt)
        if not args.no_configs:
            cfg_steps = config_block(model, dev, args.steps)

    # ---- per-kernel durations inside the step: every dispatch's own begin/end timestamps (vaa_prof_*) ----
    kern, op_us = kernel_table(recs, max(psteps, 1))
    kern_ce, op_us_ce = kernel_table(recs_ce, max(min(psteps, 4), 1))
    fused_embed = tr.embed_with is not None
    for o_ in (op_us, op_us_ce):
        if fused_embed and "K2" in o_:  # the TILED scatter + reduce belong to K2'
            o_["K2e"] = o_.get("K2e", 0.0) + o_.pop("K2")
    # the AVERAGE step of the loop: slice-only steps + 1 / INNER_LOOP of what a CE step adds (K3h's stream + finish)
    every = getattr(runner, "full_ce_every", None) if recs_ce else None
    ce_timed = sum(1 for g_ in range(args.warmup, args.warmup + args.steps) if (g_ + 1) % every == 0) if (every and run_weak) else None
    launches_slice = sum(k_["launches_per_step"] for k_ in kern.values())
    launches_avg = launches_slice + ((sum(k_["launches_per_step"] for k_ in kern_ce.values()) - launches_slice) / every if every else 0.0)
    op_us_slice = dict(op_us)
    if every:
        op_us = {o_: op_us_slice.get(o_, 0.0) + (op_us_ce.get(o_, 0.0) - op_us_slice.get(o_, 0.0)) / every for o_ in set(op_us_slice) | set(op_us_ce)}
    tfile = next((f for f in ("profiles/traffic_r06.json", "profiles/traffic_r05.json", "profiles/traffic_r04.json") if os.path.exists(os.path.join(ROOT, f))), None)
    tr_all = json.load(open(os.path.join(ROOT, tfile))) if tfile else {}
    tr_ops = tr_all.get("ops", {})

    def roofline_of(kname, label, nb, traffic, table=None):
        """the contract's roofline object for ONE hand-written kernel of the timed region (HBM-bound byte work), in-step per dispatch"""
        k = (kern if table is None else table)[kname]
        return {"kernel": f"{kname.strip('() ')} ({label})", "bound": "hbm", "achieved": nb / k["mean_us"] / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": nb / k["mean_us"] / 1e3 / HBM_PEAK_GBS, "mean_us": k["mean_us"], "min_us": k["min_us"], "samples": k["launches"],
                "algo_bytes": int(nb), "traffic": traffic}

    # the dominant kernel = the hand-written kernel the timed region actually ran with the most algorithmic bytes (and, at the 7B shape, the
    # longest): the LM head fused with K3's statistics when the step dispatches it (263 MB weight stream), else K1; K1 is reported beside it
    roofline = roofline_k1 = roofline_head = roofline_k3s = None
    k1name = next((n for n in kern if "patch_apply" in n), None)
    hname = next((n for n in kern if "head_stats_kernel" in n), None)
    sname = next((n for n in kern if "head_slice_kernel" in n), None)
    hname_ce = next((n for n in kern_ce if "head_stats_kernel" in n), None)
    if k1name:
        roofline_k1 = roofline_of(k1name, "K1", kernel_bytes(k1name, model, runner, B, patch_shape),
                                  tr_ops.get("K1t_patch_apply_fwd_tiles" if "tiles" in k1name else "K1_patch_apply_fwd", {}).get("hbm_bytes_per_launch")
                                  if (B == 64 and patch_shape == [3, 50, 50]) else None)
    if hname and kernel_bytes(hname, model, runner, B, patch_shape):
        t_head = next((v.get("hbm_bytes_per_launch") for kk, v in tr_all.items() if "head_stats_kernel" in kk and isinstance(v, dict)), None)
        if R != 128 or args.model != "openvla-7b":
            t_head = None  # the committed PMC passes ran the 7B head at 128 rows
        roofline_head = roofline_of(hname, "K3h", kernel_bytes(hname, model, runner, B, patch_shape), t_head)
    if hname_ce and not hname and kernel_bytes(hname_ce, model, runner, B, patch_shape):  # K3h runs on the CE steps only: measured there
        t_head = next((v.get("hbm_bytes_per_launch") for kk, v in tr_all.items() if "head_stats_kernel" in kk and isinstance(v, dict)), None)
        if R != 128 or args.model != "openvla-7b":
            t_head = None
        roofline_head = roofline_of(hname_ce, "K3h, on the CE steps: 1 of %d" % every, kernel_bytes(hname_ce, model, runner, B, patch_shape), t_head, table=kern_ce)
        roofline_head["launches_per_average_step"] = 1.0 / every
    if sname and kernel_bytes(sname, model, runner, B, patch_shape):
        t_k3s = next((v.get("hbm_bytes_per_launch") for kk, v in tr_all.items() if "head_slice_kernel" in kk and isinstance(v, dict)), None)
        roofline_k3s = roofline_of(sname, "K3s", kernel_bytes(sname, model, runner, B, patch_shape), t_k3s if (R == 128 and args.model == "openvla-7b") else None)
    # the dominant kernel = the hand-written kernel with the most algorithmic bytes per AVERAGE step of the loop (a kernel that runs on 1 of
    # INNER_LOOP steps counts 1 / INNER_LOOP of its bytes): K1 at the headline shape (48.2 MB against K3s's 6.3 MB and K3h's 263.7 / 50 = 5.3 MB)
    def _avg_bytes(r):
        return r["algo_bytes"] * r.get("launches_per_average_step", 1.0)
    roofline = max((r for r in (roofline_k1, roofline_head, roofline_k3s) if r), key=_avg_bytes, default=None)

    extra_full = {}
    k2_fracs = k1_standalone = copy_bw = None
    cpu_c = cpu_f = None
    cpu_thread = None
    if not args.no_cpu_baseline and world == 1:  # host-CPU leg only at N=1 (rank 0), as the bench contract asks: separate processes, started now so
        import threading                            # that they run while the GPU kernel suite below (one mostly idle host thread) does

        cpu_box = {}
        cpu_thread = threading.Thread(target=lambda: cpu_box.update(zip(("c", "f"), cpu_baseline(B, patch_shape, args.cpu_budget))), daemon=True)
        cpu_thread.start()
    if not args.no_kernel_suite and world == 1:
        del runner, model, tr  # the transform holds the model (embed_with)
        torch.cuda.empty_cache()
        from roboticattack_amd.benchmarks import device_copy_bandwidth, k1_sweep, rank_shapes

        copy_bw = device_copy_bandwidth(device=str(dev))
        ks = kernel_suite(B, patch_shape[1], patch_shape[2], device=str(dev))
        for v in ks.values():
            v["frac_of_measured_copy_bw"] = v["achieved_GBs"] / copy_bw
        k2s = k2_sweep(device=str(dev))
        extra_full.update({"measured_device_copy_GBs": copy_bw, "roofline_kernels_standalone": ks, "k2_sweep": k2s, "k1_sweep": k1_sweep(device=str(dev)),
                           "rank_shapes": rank_shapes(device=str(dev))})
        k2_fracs = [e["frac_of_8TBs"] for e in k2s]
        k1_standalone = ks.get("K1t_patch_apply_fwd_tiles", ks.get("K1_patch_apply_fwd", {})).get("achieved_GBs")
    if cpu_thread is not None:
        cpu_thread.join()
        cpu_c, cpu_f = cpu_box.get("c"), cpu_box.get("f")

    # ---------------------------------------------------------------- the records ----------------------------------------------------------------
    workload = {"uada_ddp": f"UADA_wrapper_ddp inner step: bs={B}/rank (global {B * world}), patch {args.patch}, geometry={geometry}, maskidx={maskidx or [0]}",
                "uada": f"UADA single-GPU inner step (MSE + 1/CE): bs={B}, patch {args.patch}, geometry={geometry}",
                "tma": f"TMA inner step (CE gradient): bs={B}, patch {args.patch}, geometry={geometry}",
                "upa": f"UPA inner step: bs={B}, patch {args.patch}, resize_patch={args.resize_patch}, geometry={geometry}"}[args.attack]
    config = {"workload": f"{workload}; {model_desc}; u8 frames resident in HBM",
              "global_batch": B * world, "images_per_s": (B * world * args.steps / dt) if run_weak else None, "parallelism": f"dp{world}",
              "regions": ("weak" if world == 1 else args.regions),
              "backend": (os.environ.get("VAA_DIST_BACKEND") or "nccl (RCCL)") if world > 1 else None,
              "visible_gpus": torch.cuda.device_count(), "labelled_rows_per_rank": R, "tunableop_entries_loaded_min_over_ranks": tun_min,
              "lm_head": (("K3s every step + K3h on CE steps" if sname else ("fused K3h" if hname else "GEMM + K3 rows")) if use_rows else "full logits"),
              "full_ce_every": every, "ce_steps_in_timed_region": ce_timed, "traffic_source": tfile,
              "h2d_stage_ms_per_outer_iteration": stage_ms,
              "pcie_inclusive_value_if_restaged_every_step": (world * args.steps / (dt + args.steps * stage_ms * 1e-3)) if run_weak else None}
    if world > 1:  # the collective
        config["allreduce_us_per_step"] = comm_w["mean_us_min_over_ranks"] if comm_w else None
        config["allreduce_us_per_step_max_over_ranks"] = comm_w["mean_us_max_over_ranks"] if comm_w else None
        config["comm_frac"] = comm_w["comm_frac"] if comm_w else None
        config["allreduce_back_to_back_us"] = b2b["event_us_per_call"] if b2b else None
        if strong:
            config["strong_per_rank_bs"] = strong["per_rank_bs"]
            config["strong_images_per_s"] = strong["images_per_s"]
            config["strong_ms_per_step"] = strong["ms_per_step"]
            config["strong_comm_frac"] = strong["comm_frac"]
            config["strong_allreduce_us_per_step"] = strong["allreduce_us_per_step"]["mean_us_min_over_ranks"] if strong["allreduce_us_per_step"] else None
            config["strong_speedup_vs_one_rank_weak_step"] = strong.get("speedup_vs_one_rank_weak_step")
    if per_rank:  # N=1: the per-rank ratios of the strong-scaling configs
        for b in (32, 16, 8, 4):
            pr = per_rank.get(f"bs{b}")
            if pr:
                config[f"bs{b}_images_per_s_vs_bs{B}"] = pr["images_per_s_vs_bs%d" % B]
                config[f"bs{b}_ms_per_step"] = pr["ms_per_step"]
        for n_r in (2, 4, 8):
            pr = per_rank.get(f"bs{B // n_r}")
            if pr:
                config[f"projected_strong_speedup_{n_r}_before_comm"] = pr["projected_speedup_%d_ranks_before_comm" % n_r]
    if cfg_steps:  # N=1: BASELINE configs 2 / 4 / 5, end to end
        for tag, e in cfg_steps.items():
            config[f"{tag}_ms_per_step"] = e["ms_per_step"]
            config[f"{tag}_hot_path_us"] = e["hot_path_us_per_step"]
            if e.get("dominant"):
                config[f"{tag}_dominant_kernel"] = e["dominant"]["kernel"].split("<")[0].split("(")[0].strip()
                config[f"{tag}_dominant_frac"] = e["dominant"]["frac"]
    if val_rec:  # N=1: the validation pass (f-1), un-synchronised against the per-batch read-back
        config["val_ms_per_batch_bs8"] = val_rec["ms_per_batch"]
        config["val_ms_per_batch_bs8_sync_every_batch"] = val_rec["ms_per_batch_sync_every_batch"]
    if k2_fracs:
        config["k2_sweep_frac_B64_256_1024_4096"] = k2_fracs
    if k1_standalone:
        config["k1_standalone_GBs"] = k1_standalone
    if copy_bw:
        config["measured_device_copy_GBs"] = copy_bw

    metric = "attack-steps/sec (bs=64, 3x50x50 patch, OpenVLA-7B)"
    head = {"metric": metric, "value": (world * args.steps / dt) if run_weak else None, "unit": "attack-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": (dt / args.steps * 1e3) if run_weak else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic"}
    tail = {"hot_path_us_per_step": sum(op_us.values()), "hot_path_launches_per_step": launches_avg,
            "hot_path_us_slice_step": sum(op_us_slice.values()) if every else None, "hot_path_us_ce_step": sum(op_us_ce.values()) if every else None,
            "hot_path_ops_us": op_us, "host_cpu_ms_per_step": host_cpu * 1e3 if run_weak else None,
            "peak_mem_GiB": peak_mem, "loss_finite": finite and (strong is None or strong["loss_finite_all_ranks"]),
            "profiled_pass_steps": psteps if run_weak else 0}
    wall = {"setup": t_setup, "warmup_and_timed_region": t_region, "total": time.perf_counter() - t_main}

    env_rec = {k: v for k, v in sorted(os.environ.items())
               if k.startswith(("NCCL_", "RCCL_", "HSA_", "VAA_", "TORCH_NCCL", "PYTORCH_TUNABLEOP", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE", "GPU_MAX_HW_QUEUES"))}
    full = dict(head)
    full.update({"config": dict(config, env=env_rec), "roofline": roofline, "roofline_k1": roofline_k1 if roofline is not roofline_k1 else None,
                 "roofline_head": roofline_head if roofline is not roofline_head else None, "roofline_k3s": roofline_k3s if roofline is not roofline_k3s else None,
                 "roofline_kernels": kern, "roofline_kernels_ce_steps": kern_ce, "hot_path_ops_us_slice_step": op_us_slice, "hot_path_ops_us_ce_step": op_us_ce, "strong_scaling": strong, "allreduce_us_per_step": comm_w, "allreduce_back_to_back": b2b,
                 "per_rank_step": per_rank, "config_steps": cfg_steps, "validation_pass": val_rec, "cpu_baseline": cpu_f, "host_enqueue_ms_per_step": host_enqueue * 1e3 if run_weak else None})
    full.update(tail)
    full.update(extra_full)
    full["wall_s"] = wall
    full["argv"] = sys.argv[1:]
    full_path = write_full(args.full_out, full)

    line = dict(head)
    def _short(r):  # the secondary roofline objects of the line: the figures only
        return {k_: r[k_] for k_ in ("kernel", "achieved", "frac", "mean_us", "samples", "algo_bytes", "traffic") if k_ in r} if r else None
    line.update({"config": config, "roofline": roofline, "roofline_k1": roofline_k1 if roofline is not roofline_k1 else None,
                 "roofline_head": _short(roofline_head) if roofline is not roofline_head else None,
                 "roofline_k3s": _short(roofline_k3s) if roofline is not roofline_k3s else None, "cpu_baseline": cpu_c})
    line.update(tail)
    line["wall_s"] = wall["total"]
    line["full_record"] = full_path
    if not run_weak:
        line["diagnostic"] = "regions=strong only: value not measured"
    # least important first: dropped only if the line would not fit
    optional = ["hot_path_ops_us", "config.pcie_inclusive_value_if_restaged_every_step", "config.h2d_stage_ms_per_outer_iteration", "config.visible_gpus",
                "config.bs4_ms_per_step", "config.bs8_ms_per_step", "config.bs16_ms_per_step", "config.bs32_ms_per_step", "config.measured_device_copy_GBs",
                "config.k1_standalone_GBs", "config.cfg2_hot_path_us", "config.cfg4_hot_path_us", "config.cfg5_hot_path_us", "roofline.min_us", "roofline_k1.min_us",
                "host_cpu_ms_per_step", "profiled_pass_steps", "config.images_per_s", "config.lm_head", "roofline_k3s.traffic", "roofline_head.samples",
                "roofline_k3s.samples", "hot_path_us_slice_step", "hot_path_us_ce_step"]
    try:
        out = encode_line(line, optional)
    except RuntimeError as e:  # never leave the driver without a line: fall back to the contract's keys alone (the full record has the rest)
        print(f"bench.py: {e}; printing the minimal record", file=sys.stderr)
        mini = dict(head)
        mini["config"] = {k: config.get(k) for k in ("workload", "global_batch", "parallelism", "regions", "backend")}
        mini.update({"roofline": roofline, "cpu_baseline": cpu_c, "loss_finite": tail["loss_finite"], "full_record": full_path})
        out = encode_line(mini)
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    print(out, flush=True)
    os.dup2(2, 1)  # whatever the teardown prints goes to stderr as well


if __name__ == "__main__":
    main()
