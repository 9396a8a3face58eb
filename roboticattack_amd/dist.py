"""Multi-GPU plumbing of the DDP attack (one process per GPU; `torch.distributed` backend "nccl" == RCCL over xGMI).

The reference wraps the whole 7.5 B-parameter model in DistributedDataParallel although every parameter except the
patch is frozen (UADA_ddp.py:50-51,145-150,166): DDP's constructor broadcasts ~15 GB of identical weights (C2) and each
backward all-reduces exactly one 3*ph*pw fp32 gradient plus a used-parameter bitmap (C3). Here every rank builds the
same model locally, and the only traffic per inner step is ONE all-reduce(sum) of a flat fp32 buffer
[patch gradient (30,000 B at 50x50) | 4 logging scalars]; the 1/world factor of DDP's mean is folded into K4
(`grad_scale`), so no extra elementwise kernel runs. At this size the collective is latency-bound.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def local_device() -> torch.device:
    """cuda:LOCAL_RANK — one process per GPU. Only under VAA_DIST_BACKEND=gloo (test mode: gloo moves the buffers through the host) may
    more ranks than GPUs run on a node; they then wrap around the visible devices."""
    _, _, local = env_rank_world()
    if not torch.cuda.is_available():
        return torch.device("cpu")
    n = torch.cuda.device_count()
    if os.environ.get("VAA_DIST_BACKEND") == "gloo" and n > 0:
        local %= n
    return torch.device(f"cuda:{local}")


def init_process_group(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """torchrun-style rendezvous (env://). Appendix A-D7: the reference broadcasts before initialising; here the
    group is always initialised first."""
    rank, world, local = env_rank_world()
    if not dist.is_initialized():
        if backend is None:
            # VAA_DIST_BACKEND=gloo runs the same loop with gloo moving the (GPU-resident) buffers through the host: several ranks
            # can then share ONE GPU, which is how the N > 1 path is tested on a single-GPU box (tests/test_gpu_attack.py)
            backend = os.environ.get("VAA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist.get_rank(), dist.get_world_size()


def broadcast_exp_id(exp_id: str | None) -> str:
    """UADA_wrapper_ddp.py:23-35 — rank 0's uuid for the run directory."""
    box = [exp_id]
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast_object_list(box, src=0)
    return box[0]


class PatchGradSync:
    """Flat [n_grad + n_scalars] fp32 buffer all-reduced (SUM) once per inner step (C3 + C4 fused)."""

    def __init__(self, n_grad: int, n_scalars: int, device):
        self.n_grad, self.n_scalars = n_grad, n_scalars
        self.buf = torch.zeros(n_grad + n_scalars, dtype=torch.float32, device=device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self._timing = None  # measurement aid (bench.py): [(start event, stop event, host seconds)] of every all-reduce while armed

    # ---- measurement aid: what the collective costs INSIDE a step (UADA_ddp.py:206: DDP's bucket all-reduce fires inside backward) ----
    def timing_start(self):
        """Arm: every all-reduce from here on is bracketed by two events ON THE LAUNCH STREAM (the current stream: the collective is ordered
        behind the epilogue that wrote the message, and K4 behind the collective). The bracket therefore covers the hand-off to the backend's
        stream, the wait for the slowest rank's message and the exchange itself — the time the step's critical path spends in the collective."""
        self._timing = []

    def timing_collect(self):
        """Disarm; returns [(event us, host us)] per all-reduce (event us = GPU time between the two events; host us = wall time the launching
        thread spent inside dist.all_reduce — gloo blocks the host, RCCL only enqueues)."""
        recs, self._timing = self._timing or [], None
        out = []
        for e0, e1, host_s in recs:
            e1.synchronize()
            out.append((e0.elapsed_time(e1) * 1e3, host_s * 1e6))
        return out

    def _all_reduce(self):
        if self._timing is None or not self.buf.is_cuda:
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
            return
        import time

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
        host_s = time.perf_counter() - t0
        e1.record()
        self._timing.append((e0, e1, host_s))

    def allreduce(self, grad: torch.Tensor, scalars: torch.Tensor | None = None):
        """Returns (summed grad view [n_grad], summed scalars view [n_scalars]); divide by world for means."""
        self.buf[: self.n_grad].copy_(grad.reshape(-1))
        if scalars is not None and self.n_scalars:
            self.buf[self.n_grad :].copy_(scalars.reshape(-1)[: self.n_scalars])
        if self.world > 1:
            self._all_reduce()
        return self.buf[: self.n_grad], self.buf[self.n_grad :]

    def allreduce_step(self, grad: torch.Tensor, loss_scalars: torch.Tensor, pick: torch.Tensor):
        """One inner step's message [grad | loss_scalars[pick]] packed with two small launches (copy, gather), then ONE all-reduce(sum).
        `pick` is an int64 device index of n_scalars entries. (The fused step writes the same message with vaa_step_epilogue and calls
        allreduce_packed.)"""
        g = grad.reshape(-1)
        self.buf[: self.n_grad].copy_(g)
        torch.index_select(loss_scalars, 0, pick, out=self.buf[self.n_grad : self.n_grad + pick.numel()])
        return self.allreduce_packed()

    def allreduce_packed(self):
        """The message is already in `buf` ([grad | scalars], e.g. written by vaa_step_epilogue): ONE all-reduce(sum), views returned."""
        if self.world > 1:
            self._all_reduce()
        return self.buf[: self.n_grad], self.buf[self.n_grad :]


def broadcast_patch(patch: torch.Tensor, src: int = 0) -> torch.Tensor:
    """UADA_ddp.py:140-144 (C1)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(patch, src=src)
    return patch


def allreduce_scalar(value: float, op: str, device) -> float:
    """UADA_ddp.py:214-221, 275-280 (C4/C5): AVG or MAX of one logging scalar."""
    t = torch.tensor([value], dtype=torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if op == "MAX":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t /= dist.get_world_size()
    return float(t.item())
