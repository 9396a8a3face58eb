#!/usr/bin/env python3
"""bench.py — attack-steps/sec of the UADA inner loop (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one inner-loop iteration of UADA_wrapper_ddp (UADA_ddp.py:189-209):
    host RNG draws -> K1 paste/warp (HIP) -> OpenVLA-7B-shaped bf16 forward + activation backward (PyTorch-ROCm)
    -> K3 loss fwd+bwd on the labelled rows (HIP) -> K2' patch-embed backward on the tiles under the patch (MFMA) + patch-grad gather
    (HIP; plain K2 on the pixel gradient with VAA_FUSED_EMBED_GRAD=0) -> [RCCL all-reduce, 30 KB] -> K4 AdamW+clamp (HIP)
Workload: bs=64 PER RANK (reference semantics, UADA_ddp.py:158 -> weak scaling), 3x50x50 patch, geometry=True, maskidx=[0],
synthetic BridgeData-shaped frames resident in HBM as u8, random-init weights of the OpenVLA-7B architecture.
Prints ONE JSON line on rank 0. The timed region (`value`) runs UN-profiled; a short second pass of the same steps then runs with the
library's per-dispatch timer armed (vaa_prof_*: every hand-written kernel launched through hipExtLaunchKernel with its own start/stop event
pair = that dispatch's begin/end timestamps, no marker brackets, no subtraction) and gives `roofline` — the dominant hand-written kernel of
the path IN-STEP: `head_stats_kernel`, the LM head fused with K3's statistics, ONE pass over the 263 MB head weight (264.8 of ~334 MB of the
path's algorithmic bytes per step, and its longest kernel); K1's `patch_apply_tiles_kernel` (48.2 MB) is `roofline_k1`; the back-to-back
figure is reported next to it as standalone_* — plus `roofline_kernels` / `hot_path_ops`. `k2_sweep` carries the K2 batch sweep; `cpu_baseline`
is the reference's PyTorch-CPU op sequence for the same replaced ops (oracle/ref_port.py) timed on this box's host cores.
N > 1 (one rank per GPU, RCCL): both timed regions — weak (`value`, bs per rank) and `strong_scaling` (the same global batch split over the
ranks) — carry the collective's in-step cost (`allreduce_us_per_step`, `comm_frac`, measured with events on the launch stream around the
30 KB all-reduce) and `config.env` records the NCCL_* / RCCL_* / HSA_* / VAA_* environment of the run. `--regions weak|strong|both` selects the
regions (a full-size 8-rank functional run on ONE GPU fits with `--regions strong`).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bs", type=int, default=64, help="per-rank batch (UADA_ddp.py:158)")
    ap.add_argument("--patch", type=str, default="3,50,50")
    ap.add_argument("--model", type=str, default="openvla-7b", choices=["openvla-7b", "tiny", "surrogate"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-suite", action="store_true")
    ap.add_argument("--no-per-rank", action="store_true", help="skip the bs=8 / bs=4 per-rank step block of the N=1 record")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of host CPU work for cpu_baseline")
    ap.add_argument("--regions", type=str, default="both", choices=["weak", "strong", "both"],
                    help="N > 1: which timed regions run (weak = bs per rank, the headline; strong = the global batch split over the ranks). "
                         "`strong` alone is a functional / diagnostic run: `value` is then null")
    ap.add_argument("--profile-steps", type=int, default=-1, help="steps of the separate per-dispatch-profiled pass (-1: min(steps, 10); 0: none)")
    return ap.parse_args()


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
        return build_openvla(openvla_7b_cfg(), device=dev, dtype=torch.bfloat16, seed=0), "OpenVLA-7B shape (DINOv2-L/14 + SigLIP-so400m/14 + Llama-2-7B), random init"
    if kind == "tiny":
        return build_openvla(tiny_cfg(), device=dev, dtype=torch.bfloat16, seed=0), "tiny topology-equal model (bf16)"
    from roboticattack_amd.surrogate import SurrogateVLA

    return SurrogateVLA(seed=0).to(dev), "fp32 surrogate"


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
    and with ALL logical cores: 2 warm-ups, then 5 timed iterations, min and median reported.

    Bounded: every leg times a slice of the per-rank batch whose size a 2-image probe picks so that 2 + 5 iterations fit the leg's
    budget (one K3 iteration at bs=64 on one thread takes ~30 s) and scales linearly to bs; every leg runs in its own process under a
    wall-clock guard, because torch.set_num_threads(256) thrashes on the reference's many tiny per-image ops (a single-image iteration
    was seen to take seconds): a leg that exceeds its guard is reported as timed out. Hosts with more than 32 logical cores also get a
    32-thread leg (the setting the CPU path ran best at in round 1)."""
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
    S = ok[best]["logits_S"] if best else 0
    return {
        "value": ok[best]["steps_per_s_min_time"] if best else None, "unit": "patch-path steps/s on host CPU (K1+K2+K3+K4 only, model excluded)",
        "cores": ok[best]["threads"] if best else 0, "kind": "port",
        "sample": f"oracle/ref_port.py (PyTorch-CPU restatement of the reference op sequence), patch {patch_shape}, geometry=True, fp32 logits "
                  f"[bs,{S},32064]; legs: 1 thread, all {ncores} logical cores" + (", 32 threads" if ncores > 32 else "") + "; per leg 2 warm-ups "
                  f"then 5 timed iterations, min and median, on a slice of the bs={bs} batch sized by a 2-image probe (sample_bs_* in `legs`), "
                  f"scaled linearly to bs={bs}; each leg in its own process under a wall-clock guard; value = the fastest leg's min",
        "legs": legs, "host_logical_cores": ncores,
    }


class StepRunner:
    """One rank's attack-loop state for a per-rank batch of B images: the inner step of attack/uada_ddp.py — the SAME code
    (AttackBase.fused_ddp_step / model_loss) — on synthetic frames resident in HBM."""

    def __init__(self, model, dev, B, patch_shape, rank, world):
        from roboticattack_amd import dist as vdist
        from roboticattack_amd import ops, synthetic
        from roboticattack_amd.attack.engine import AttackBase
        from roboticattack_amd.labels import mask_labels
        from roboticattack_amd.optim import PatchOptimizer

        self.ops, self.model, self.dev, self.B, self.world = ops, model, dev, B, world
        self.att = AttackBase(model, None, "", "adamW", False)  # K2'  when the model exposes its patch-embed weights; VAA_FUSED_EMBED_GRAD=0: plain K2
        self.use_rows = self.att.use_rows
        self.tr = self.att.randomPatchTransform
        self.fused = self.att.fused_ddp_available()  # K2's final sum + K3's fold + the DDP message in one launch (VAA_FUSED_EPILOGUE=0: separate launches)
        self.batch = synthetic.synth_batch(1234 + rank, B, "noise", as_pil=False)
        self.img = self.tr.stage_images(torch.from_numpy(self.batch["pixel_values"]))
        self.input_ids = self.batch["input_ids"].to(dev)
        self.attn = self.batch["attention_mask"].to(dev)
        self.labels = mask_labels(self.batch["labels"].clone(), [0]).to(dev)
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

    def step(self):  # attack/uada_ddp.py inner step
        a, ops = self.att, self.ops
        self.opt.zero_grad()
        if self.fused and self.world == 1:
            # host draws -> K1 (tile-major) -> model -> K3 statistics -> backward -> K2' tiles + scatter -> epilogue incl. K4 (nothing to exchange)
            a.fused_ddp_step(self.img, self.patch, self.input_ids, self.attn, self.labels, True, 5.0, self.sync.buf, self.scal, optimizer=self.opt)
            return
        if self.fused:
            # ... -> epilogue: the message is in sync.buf -> all-reduce -> K4
            a.fused_ddp_step(self.img, self.patch, self.input_ids, self.attn, self.labels, True, 5.0, self.sync.buf, self.scal)
            g_sum, _ = self.sync.allreduce_packed()  # [grad | CE, MSE, UAD, total]: one all-reduce per step
        else:
            pix = self.tr.apply_random_patch_batch(self.img, self.patch, mean=a.mean, std=a.std, geometry=True)  # host RNG draws + K1
            total, scalars, _ = a.model_loss(self.input_ids, self.attn, pix, self.labels, ops.LOSS_UADA_DDP, w=5.0)  # model + LM head + K3
            total.backward()  # ... -> K2 (or K2' fed by the patch-embed output gradients)
            g_sum, _ = self.sync.allreduce_step(self.patch.grad, scalars, self.pick)
            self.scal.copy_(scalars)
        self.opt.step(grad=g_sum.view_as(self.patch), grad_scale=self.inv_world)  # K4


def comm_summary(recs, world, steps, dt, dev):
    """In-step cost of the gradient exchange from the per-call event brackets of PatchGradSync (one all-reduce per step): rank 0's own figures
    plus min / max over ranks of the per-rank means — the rank that reaches the collective LAST sees the exchange alone, the others also wait
    for it (skew), so min ~ hand-off + exchange latency and max - min ~ rank skew."""
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
            "comm_frac": mean_min / step_us, "comm_plus_skew_frac": mean_max / step_us,
            "note": "events on the launch stream around dist.all_reduce of the [grad | 4 scalars] message (30,016 B at 50x50), every timed step; "
                    "comm_frac = min-over-ranks mean / step time (the exchange itself on the critical path), comm_plus_skew_frac = max-over-ranks mean / step time "
                    "(adds the wait for the slowest rank's message)"}


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
    sync.buf.zero_()  # n sums of a non-zero message can overflow to inf; the next step rewrites the buffer anyway
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
KERNEL_OPS = (("patch_apply_fwd_kernel", "K1"), ("patch_apply_tiles_kernel", "K1"), ("embed_dgrad_tiles", "K2e"), ("patch_grad_scatter_kernel", "K2"), ("patch_grad_reduce_kernel", "K2"),
              ("head_stats_kernel", "K3h"), ("head_finish_kernel", "K3h"), ("rows_stats_kernel", "K3"), ("rows_finish_kernel", "K3"), ("loss_stats_kernel", "K3"), ("loss_grad_kernel", "K3"), ("step_epilogue_kernel", "EPI"), ("patch_update_kernel", "K4"),
              ("patch_resize", "K0"))


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
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    import torch.distributed as dist

    from roboticattack_amd import dist as vdist
    from roboticattack_amd import ops
    from roboticattack_amd.benchmarks import HBM_PEAK_GBS, algo_bytes, k2_sweep, kernel_suite

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
    model, model_desc = build_model(args.model, dev)
    runner = StepRunner(model, dev, B, patch_shape, rank, world)
    use_rows, tr, R = runner.use_rows, runner.tr, runner.R

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
    recs, comm_w, finite = [], None, True
    if run_weak:
        dt, host_enqueue, host_cpu, _, crecs = timed_steps(runner, args.steps, args.warmup, world, dev, comm=True)
        comm_w = comm_summary(crecs, world, args.steps, dt, dev)
        finite = bool(torch.isfinite(runner.scal).all())
        # ---- the same steps once more, SEPARATELY, with the library's per-dispatch timer armed: the in-step kernel durations ----
        if psteps > 0:
            _, _, _, recs, _ = timed_steps(runner, psteps, 1, world, dev, profile=True)
    t_region = time.perf_counter() - t_main - t_setup
    peak_mem = torch.cuda.max_memory_allocated(dev) / 2**30
    b2b = allreduce_back_to_back(runner.sync, world) if world > 1 else None

    # ---- strong scaling (BASELINE config 3: "bs=64 over N GPUs"): the same global batch split over the ranks ----
    strong = None
    if run_strong:
        bs_s = max(1, B // world)
        torch.cuda.empty_cache()  # the weak region's cached activation blocks are not needed any more (ranks that share a GPU in test mode are near its capacity)
        r_s = StepRunner(model, dev, bs_s, patch_shape, rank, world)
        dt_s, enq_s, cpu_s, _, crecs_s = timed_steps(r_s, args.steps, args.warmup, world, dev, comm=True)
        comm_s = comm_summary(crecs_s, world, args.steps, dt_s, dev)
        fin_s = torch.tensor([1.0 if bool(torch.isfinite(r_s.scal).all()) else 0.0], device=dev)
        dist.all_reduce(fin_s, op=dist.ReduceOp.MIN)
        mem_s = torch.tensor([torch.cuda.max_memory_allocated(dev) / 2**30], dtype=torch.float64, device=dev)
        dist.all_reduce(mem_s, op=dist.ReduceOp.MAX)
        strong = {"per_rank_bs": bs_s, "global_batch": bs_s * world, "ms_per_step": dt_s / args.steps * 1e3, "steps_per_s": args.steps / dt_s,
                  "images_per_s": bs_s * world * args.steps / dt_s, "host_cpu_ms_per_step": cpu_s * 1e3, "host_enqueue_ms_per_step": enq_s * 1e3,
                  "loss_finite_all_ranks": bool(fin_s.item() > 0.5), "peak_mem_GiB_max_over_ranks": float(mem_s.item()),
                  "allreduce_us_per_step": comm_s, "comm_frac": comm_s["comm_frac"] if comm_s else None,
                  "note": "global batch fixed at the N=1 workload's bs, split evenly over the ranks (BASELINE config 3); timed like the weak region: "
                          "W warm-up steps, K steps between barrier + synchronize, max over ranks; un-profiled"}
        if run_weak:
            strong["speedup_vs_one_rank_weak_step"] = (bs_s * world * args.steps / dt_s) / (B * args.steps / dt)
            strong["speedup_note"] = ("images/s of the split global batch over ALL ranks / images/s of ONE rank's bs=%d step in this same run (the weak region's "
                                      "per-rank rate): the strong-scaling speedup 1 -> %d ranks measured inside one job" % (B, world))
        del r_s
    # every rank's hipBLASLt / rocBLAS selections (a rank that lost them runs ~6 % slower and drags the synchronous step): min over ranks
    tun = _tunable_entries()
    tun_min = tun
    if world > 1:
        tt = torch.tensor([float(tun)], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        tun_min = int(tt.item())

    # ---- the per-rank step of the strong-scaling configs on ONE GPU (N=1 only): what each of 8 ranks runs in configs 3/4 (bs=8) and 5 (bs=4) ----
    per_rank = None
    if world == 1 and not args.no_per_rank and args.model == "openvla-7b":
        per_rank = {}
        cfg_m = getattr(model, "cfg", None)
        ips64 = B * args.steps / dt
        for b in (32, 16, 8, 4):
            torch.cuda.empty_cache()
            rb = StepRunner(model, dev, b, patch_shape, rank, world)
            n_b = max(args.steps, 10) if b <= 8 else max(args.steps // 2, 6)
            dt_b, enq_b, cpu_b, _, _ = timed_steps(rb, n_b, 3, world, dev)  # un-profiled, like the headline it is compared with
            n_p = min(n_b, 5)
            _, _, _, recs_b, _ = timed_steps(rb, n_p, 0, world, dev, profile=True)
            hot_b = {}
            for name, us in recs_b:  # the hand-written launches of the step at this batch, per dispatch (vaa_prof_*)
                op = next((o for sub, o in KERNEL_OPS if sub in name), "other")
                hot_b[op] = hot_b.get(op, 0.0) + us / n_p
            head_us = [us for name, us in recs_b if "head_stats_kernel" in name]
            per_rank[f"bs{b}"] = {"ms_per_step": dt_b / n_b * 1e3, "images_per_s": b * n_b / dt_b, "images_per_s_vs_bs%d" % B: (b * n_b / dt_b) / ips64,
                                  "projected_speedup_%d_ranks_before_comm" % (B // b): (B // b) * (b * n_b / dt_b) / ips64,
                                  "host_cpu_ms_per_step": cpu_b * 1e3, "host_enqueue_ms_per_step": enq_b * 1e3, "labelled_rows": rb.R,
                                  "hot_path_us_per_step": sum(hot_b.values()), "hot_path_ops_us": hot_b}
            if head_us and cfg_m is not None:  # the fused LM head (a 263 MB weight stream at the 7B shape): in-step duration and HBM rate
                hb = 2 * 32064 * cfg_m.llm_dim
                per_rank[f"bs{b}"]["fused_head"] = {"kernel": "head_stats_kernel", "mean_us": float(np.mean(head_us)), "algo_bytes": hb,
                                                    "achieved_GBs": hb / float(np.mean(head_us)) / 1e3, "frac": hb / float(np.mean(head_us)) / 1e3 / HBM_PEAK_GBS}
            del rb
        per_rank["note"] = ("full-model step (same code path as the timed region) at the per-rank batches of the strong-scaling runs — global 64 over 2 / 4 / 8 "
                            "ranks -> bs = 32 / 16 / 8 (BASELINE config 3; config 4: 32 over 4 -> 8), config 5: 32 over 8 -> bs=4 — on this one GPU; projected "
                            "speedup = ranks x images/s at that batch / images/s at bs=%d, i.e. the strong-scaling ceiling before the 30 KB all-reduce" % B)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel durations inside the timed region: every dispatch's own begin/end timestamps (vaa_prof_*) ----
    per = {}
    for name, us in recs:
        per.setdefault(name, []).append(us)
    esz = 2 if use_rows and args.model == "openvla-7b" else 4
    cfg = getattr(model, "cfg", None)
    embed_width = (cfg.dino.dim + cfg.siglip.dim) if cfg is not None and hasattr(cfg, "dino") else 1024 + 1152  # K2': dY row width of both towers
    kern, op_us = {}, {}
    for name, ts in per.items():
        op = next((o for sub, o in KERNEL_OPS if sub in name), "other")
        t = np.asarray(ts)
        kern[name] = {"op": op, "launches": len(ts), "launches_per_step": len(ts) / psteps, "mean_us": float(t.mean()), "median_us": float(np.median(t)),
                      "min_us": float(t.min()), "max_us": float(t.max())}
        op_us[op] = op_us.get(op, 0.0) + float(t.sum()) / psteps
    fused = tr.embed_with is not None
    if fused and "K2" in op_us:  # the TILED scatter + reduce belong to K2'
        op_us["K2e"] = op_us.get("K2e", 0.0) + op_us.pop("K2")
    op_bytes = {"K1": algo_bytes("K1", B, patch_shape[1], patch_shape[2]), "K2": algo_bytes("K2", B, patch_shape[1], patch_shape[2]),
                "K2e": algo_bytes("K2e", B, patch_shape[1], patch_shape[2], embed_width=embed_width),
                "K3": algo_bytes("K3_slice" if use_rows else "K3", B, rows=R, esize=esz), "K4": algo_bytes("K4", B, patch_shape[1], patch_shape[2])}
    if "K3h" in op_us and cfg is not None and hasattr(cfg, "llm_dim"):  # LM head fused with K3's statistics: the head weight streamed once + the hidden rows
        op_bytes["K3h"] = 2 * 32064 * cfg.llm_dim + 2 * R * cfg.llm_dim
    hot_ops = {o: {"us_per_step": u, "algo_bytes": op_bytes.get(o), "achieved_GBs": (op_bytes[o] / u / 1e3 if o in op_bytes else None),
                   "frac": (op_bytes[o] / u / 1e3 / HBM_PEAK_GBS if o in op_bytes else None)} for o, u in op_us.items()}
    k1name = next((n for n in kern if "patch_apply_tiles_kernel" in n or "patch_apply_fwd_kernel" in n), None)
    hname = next((n for n in kern if "head_stats_kernel" in n), None)
    tfile = next((f for f in ("profiles/traffic_r04.json", "profiles/traffic_r03.json", "profiles/traffic_r02.json") if os.path.exists(os.path.join(ROOT, f))), None)
    tr_all = json.load(open(os.path.join(ROOT, tfile))) if tfile else {}
    tr_ops = tr_all.get("ops", {})

    def roofline_of(kname, label, nb, traffic):
        """the contract's roofline object for ONE hand-written kernel of the timed region (HBM-bound byte work)"""
        k = kern[kname]
        r = {"kernel": f"{kname} ({label})", "bound": "hbm", "achieved": nb / k["mean_us"] / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": nb / k["mean_us"] / 1e3 / HBM_PEAK_GBS, "mean_us": k["mean_us"], "min_us": k["min_us"], "samples": k["launches"],
             "algo_bytes": nb, "traffic": traffic,
             "traffic_source": (tfile or "none") + " (builder-side rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated on a 512 MiB copy in the same passes; not re-measured by this run)",
             "timing": "IN-STEP, per dispatch: after the (un-profiled) timed region the same steps run once more with the library's per-dispatch timer armed — "
                       "every launch of the library goes through hipExtLaunchKernel with its own start/stop event pair, which the runtime binds to that dispatch's "
                       "begin/end timestamps, the quantity rocprofv3 --kernel-trace reports (profiles/r04_bench_kernel_stats.csv is the rocprofv3 summary of "
                       "the same command); mean over all launches of that pass, no marker brackets, no subtraction"}
        # builder-side cross reference (NOT measured by this run): rocprofv3's average for the same kernel in the committed summary of the same command
        ref = os.path.join(ROOT, "profiles", "r04_bench_kernel_stats.csv")
        if os.path.exists(ref):
            import csv

            for row in csv.DictReader(open(ref)):
                if kname.strip("() ").split("(")[0] in row["Name"]:  # ("(head_stats_kernel<4>)" / "patch_apply_tiles_kernel": the launch site's spelling)
                    us = float(row["AverageNs"]) * 1e-3
                    r["rocprofv3_reference"] = {"file": "profiles/r04_bench_kernel_stats.csv", "calls": int(row["Calls"]), "mean_us": us,
                                                "frac": nb / us / 1e3 / HBM_PEAK_GBS,
                                                "note": "committed rocprofv3 --kernel-trace --stats summary of `bench.py --steps 20 --warmup 3 --no-cpu-baseline "
                                                        "--no-kernel-suite --no-per-rank`; the per-dispatch events of an un-profiled run read 0.3-1.8 us above it "
                                                        "(they include the dispatch's start-up after the preceding command), so the line's frac is the lower one"}
                    break
        return r

    # the dominant kernel = the hand-written kernel of the timed region with the most algorithmic bytes (and, at the 7B shape, the longest): the
    # LM head fused with K3's statistics when the step runs it (263 MB weight stream), else K1; the other one is reported beside it
    roofline = roofline_k1 = roofline_head = None
    if k1name:
        roofline_k1 = roofline_of(k1name, "K1", op_bytes["K1"],
                                  tr_ops.get("K1t_patch_apply_fwd_tiles" if "tiles" in k1name else "K1_patch_apply_fwd", {}).get("hbm_bytes_per_launch"))
    if hname and "K3h" in op_bytes:
        t_head = next((v.get("hbm_bytes_per_launch") for kk, v in tr_all.items() if "head_stats_kernel" in kk and isinstance(v, dict)), None)
        if R != 128 or args.model != "openvla-7b":
            t_head = None  # the committed PMC passes ran the 7B head at 128 rows
        roofline_head = roofline_of(hname, "K3h: LM head + K3 statistics, one pass over the head weight", op_bytes["K3h"], t_head)
    roofline = max((r for r in (roofline_k1, roofline_head) if r), key=lambda r: r["algo_bytes"], default=None)
    if roofline:
        by = {o: b for o, b in op_bytes.items() if o in op_us and b}
        roofline["note"] = ("dominant = the hand-written kernel of the timed region with the most algorithmic bytes (%.1f of ~%.0f MB per step); every hand-written "
                            "kernel of the timed region is listed in roofline_kernels, per-operator sums in hot_path_ops, K1 in roofline_k1"
                            % (roofline["algo_bytes"] / 1e6, sum(by.values()) / 1e6))

    extra = {"host_enqueue_ms_per_step": host_enqueue * 1e3 if run_weak else None, "host_cpu_ms_per_step": host_cpu * 1e3 if run_weak else None,
             "host_overhead_note": "host_cpu = CPU time of the launching thread per step (python + HIP launches): what N ranks on one host need N times in parallel; "
                                   "host_enqueue = wall time inside step() without an explicit sync (it also contains waits on a full launch queue / staged H2D copies, "
                                   "so it scales with the GPU work); the step is GPU-bound while ms_per_step exceeds host_cpu",
             "hot_path_us_per_step": sum(op_us.values()), "hot_path_launches_per_step": sum(k["launches_per_step"] for k in kern.values())}
    if not args.no_kernel_suite:
        del runner, model, tr  # the transform holds the model (embed_with)
        torch.cuda.empty_cache()
        from roboticattack_amd.benchmarks import device_copy_bandwidth, k1_sweep, rank_shapes

        copy_bw = device_copy_bandwidth(device=str(dev))
        extra["measured_device_copy_GBs"] = copy_bw
        extra["roofline_kernels_standalone"] = kernel_suite(B, patch_shape[1], patch_shape[2], device=str(dev))
        for v in extra["roofline_kernels_standalone"].values():
            v["frac_of_measured_copy_bw"] = v["achieved_GBs"] / copy_bw
        extra["k2_sweep"] = k2_sweep(device=str(dev))
        extra["k1_sweep"] = k1_sweep(device=str(dev))
        extra["rank_shapes"] = rank_shapes(device=str(dev))
        ks = extra["roofline_kernels_standalone"]
        used_k2 = "K2e_patch_embed_grad_gather" if fused else "K2_patch_grad_gather"
        gpu_ops_s = sum(v["mean_us"] for k, v in ks.items() if (not k.startswith("K2") or k == used_k2) and k not in ("K3_full_rows_fwd_bwd", "K3_full_one_launch_optin", "K3h_head_loss_rows_stats", "K3h_gemm_path_for_comparison")) * 1e-6
        extra["gpu_patch_path_steps_per_s"] = 1.0 / gpu_ops_s
        # the same kernels launched back to back (hipGraph replays of 10 launches between two events, same process) next to the in-step figures
        for r, key in ((roofline_k1, "K1_patch_apply_fwd"), (roofline_head, "K3h_head_loss_rows_stats")):
            if r and key in ks:
                k = ks[key]
                r.update({"measured_device_copy_GBs": copy_bw, "standalone_mean_us": k["mean_us"], "standalone_achieved": k["achieved_GBs"],
                          "standalone_frac": k["achieved_GBs"] / HBM_PEAK_GBS, "frac_of_measured_copy_bw": r["achieved"] / copy_bw,
                          "standalone_frac_of_measured_copy_bw": k["achieved_GBs"] / copy_bw})
    cpu = None
    if not args.no_cpu_baseline and world == 1:  # host-CPU leg only at N=1 (rank 0), as the bench contract asks
        cpu = cpu_baseline(B, patch_shape, args.cpu_budget)

    env_rec = {k: v for k, v in sorted(os.environ.items())
               if k.startswith(("NCCL_", "RCCL_", "HSA_", "VAA_", "TORCH_NCCL", "PYTORCH_TUNABLEOP", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE", "GPU_MAX_HW_QUEUES"))}
    config = {"workload": f"UADA_wrapper_ddp inner step: bs={B} per rank (global {B * world}), patch {args.patch}, geometry=True, maskidx=[0], "
                          f"{model_desc}; frames resident in HBM as u8",
              "global_batch": B * world, "images_per_s": (B * world * args.steps / dt) if run_weak else None, "parallelism": f"dp{world}",
              "regions": ("weak" if world == 1 else args.regions),
              "backend": (os.environ.get("VAA_DIST_BACKEND") or "nccl (RCCL)") if world > 1 else None,
              "visible_gpus": torch.cuda.device_count(),
              "labelled_rows_per_rank": R, "tunableop_entries_loaded": tun, "tunableop_entries_loaded_min_over_ranks": tun_min,
              "lm_head": "labelled rows only" if use_rows else "full logits",
              "h2d_stage_ms_per_outer_iteration": stage_ms,
              "pcie_inclusive_value_if_restaged_every_step": (world * args.steps / (dt + args.steps * stage_ms * 1e-3)) if run_weak else None,
              "env": env_rec}
    if world > 1:  # the collective, where `parsed.config` keeps it
        config["allreduce_us_per_step"] = comm_w["mean_us_min_over_ranks"] if comm_w else None
        config["allreduce_us_per_step_max_over_ranks"] = comm_w["mean_us_max_over_ranks"] if comm_w else None
        config["comm_frac"] = comm_w["comm_frac"] if comm_w else None
        config["allreduce_back_to_back_us"] = b2b["event_us_per_call"] if b2b else None
        if strong:
            config["strong_images_per_s"] = strong["images_per_s"]
            config["strong_ms_per_step"] = strong["ms_per_step"]
            config["strong_comm_frac"] = strong["comm_frac"]
            config["strong_speedup_vs_one_rank_weak_step"] = strong.get("speedup_vs_one_rank_weak_step")
    if per_rank:  # N=1: the per-rank ratios of the strong-scaling configs, where `parsed.config` keeps them
        for b in (32, 16, 8, 4):
            pr = per_rank.get(f"bs{b}")
            if pr:
                config[f"bs{b}_images_per_s_vs_bs{B}"] = pr["images_per_s_vs_bs%d" % B]
                config[f"bs{b}_ms_per_step"] = pr["ms_per_step"]
        for n_r in (2, 4, 8):
            pr = per_rank.get(f"bs{B // n_r}")
            if pr:
                config[f"projected_strong_speedup_{n_r}_before_comm"] = pr["projected_speedup_%d_ranks_before_comm" % n_r]
    line = {
        "metric": "attack-steps/sec (bs=64, 3x50x50 patch, OpenVLA-7B)", "value": (world * args.steps / dt) if run_weak else None,
        "unit": "attack-steps/s (one unit = one bs-64 inner step on one rank; whole job = ranks x synchronous steps)",
        "sync_steps_per_s": (args.steps / dt) if run_weak else None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": (dt / args.steps * 1e3) if run_weak else None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": config,
        "roofline": roofline, "roofline_k1": roofline_k1 if roofline is not roofline_k1 else None, "strong_scaling": strong, "allreduce_us_per_step": comm_w, "comm_frac": comm_w["comm_frac"] if comm_w else None,
        "allreduce_back_to_back": b2b, "per_rank_step": per_rank, "hot_path_ops": hot_ops, "roofline_kernels": kern, "cpu_baseline": cpu,
        "peak_mem_GiB": peak_mem, "loss_finite": finite and (strong is None or strong["loss_finite_all_ranks"]),
        "profiled_pass_steps": psteps if run_weak else 0,
    }
    if not run_weak:
        line["note"] = "--regions strong: a functional / diagnostic run of the strong-scaling region only; `value` (the weak-scaling headline) was not measured"
    line.update(extra)
    # where this process's wall time went (imports excluded): model + batch set-up, warm-up + timed region, everything reported beside it
    line["wall_s"] = {"setup": t_setup, "warmup_and_timed_region": t_region, "total": time.perf_counter() - t_main}
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    print(json.dumps(line), flush=True)
    os.dup2(2, 1)  # whatever the teardown prints goes to stderr as well
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
