"""world_size-2 `gloo` test of the multi-GPU plumbing on CPU (roboticattack_amd/dist.py): rendezvous on 127.0.0.1,
initial patch broadcast (C1), run-id broadcast (C6), fused grad+scalar all-reduce (C3+C4), scalar AVG/MAX (C5), and the
data-parallel identity the design relies on: mean over ranks of per-shard patch gradients == gradient of the full batch."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from oracle import c_oracle
    from roboticattack_amd import dist as vdist
    from roboticattack_amd import synthetic
    from roboticattack_amd.benchmarks import random_params

    r, w = vdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    # C1 / C6
    patch = torch.rand(3, 50, 50) if rank == 0 else torch.empty(3, 50, 50)
    vdist.broadcast_patch(patch)
    exp_id = vdist.broadcast_exp_id("run-abc" if rank == 0 else None)
    # every rank holds the same patch; its shard of a global batch of 4 images
    B, per = 4, 4 // world
    imgs = synthetic.synth_images(5, B, "smooth")
    xy, th = random_params(B, 50, 50, 42)
    g = synthetic.synth_upstream_grad(9, B)
    gb = g.view(torch.int16).numpy().view(np.uint16)
    sl = slice(rank * per, (rank + 1) * per)
    # per-rank loss is a mean over LOCAL rows -> local upstream grad carries 1/per; the global mean carries 1/B
    local = c_oracle.patch_grad(gb[sl], patch.numpy(), xy[sl], th[sl], 1, 0) / per
    sync = vdist.PatchGradSync(7500, 4, dev)
    g_sum, s_sum = sync.allreduce(torch.from_numpy(local), torch.tensor([1.0 + rank, 2.0, 3.0 * rank, 4.0]))
    mean_grad = (g_sum / world).numpy().reshape(3, 50, 50).copy()
    mx = vdist.allreduce_scalar(float(rank), "MAX", dev)
    av = vdist.allreduce_scalar(float(rank), "AVG", dev)
    full = c_oracle.patch_grad(gb, patch.numpy(), xy, th, 1, 0) / B
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), patch=patch.numpy(), mean_grad=mean_grad, full=full, s=s_sum.numpy(), mx=mx, av=av,
             exp=np.array(exp_id))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["patch"], r1["patch"])  # C1
    assert str(r0["exp"]) == str(r1["exp"]) == "run-abc"  # C6
    assert np.array_equal(r0["mean_grad"], r1["mean_grad"]), "all ranks must apply bit-identical averaged gradients"
    assert np.abs(r0["mean_grad"] - r0["full"]).max() <= 2e-6 * np.abs(r0["full"]).max()  # N ranks == 1 rank with N x batch
    assert np.allclose(r0["s"], [3.0, 4.0, 3.0, 8.0]) and np.array_equal(r0["s"], r1["s"])  # fused scalars (sums)
    assert r0["mx"] == r1["mx"] == 1.0 and r0["av"] == r1["av"] == 0.5


def test_local_device_follows_local_rank(monkeypatch):
    """dist.local_device(): cuda:LOCAL_RANK, wrapping around the visible GPUs only in the gloo test mode; the CPU when there is no GPU."""
    import torch

    from roboticattack_amd import dist as vdist

    monkeypatch.setenv("RANK", "5")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert vdist.env_rank_world() == (5, 8, 5)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    assert vdist.local_device() == torch.device("cpu")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.delenv("VAA_DIST_BACKEND", raising=False)
    assert vdist.local_device() == torch.device("cuda:5")  # one process per GPU: no silent wrap-around in production
    monkeypatch.setenv("VAA_DIST_BACKEND", "gloo")
    assert vdist.local_device() == torch.device("cuda:1")



def _finite_worker(rank, world, port, out_dir):
    import sys
    import types

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from roboticattack_amd import dist as vdist
    from roboticattack_amd.attack.engine import AttackBase, NonFiniteAttackState

    vdist.init_process_group("gloo")
    patch = torch.rand(3, 8, 8)
    opt = types.SimpleNamespace(m=torch.zeros(3, 8, 8), v=torch.zeros(3, 8, 8), last_stats=torch.zeros(2))
    res = []
    # round 0: every rank finite -> nobody raises; round 1: rank 1's moments are poisoned -> BOTH ranks raise, and rank 0 says it was not its own state
    for rnd in range(2):
        if rnd == 1 and rank == 1:
            opt.m[0, 0, 0] = float("nan")
        try:
            AttackBase.assert_finite_state(None, patch, opt, np.zeros((2, 8)), f"round {rnd}", all_ranks=True)
            res.append("ok")
        except NonFiniteAttackState as e:
            res.append("raised-local" if "detected on THIS rank" in str(e) else "raised-remote")
    with open(os.path.join(out_dir, f"f{rank}.txt"), "w") as f:
        f.write(",".join(res))
    dist.barrier()  # both ranks get here: nobody was left waiting in a collective
    dist.destroy_process_group()


def test_finite_state_verdict_is_rank_consistent(tmp_path):
    """in the data-parallel loop the non-finite verdict is all-reduced (MIN) before anyone raises — a rank whose own state is fine
    raises together with the rank that saw the NaN instead of blocking in the next all-reduce until the collective times out."""
    world = 2
    mp.spawn(_finite_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "f0.txt").read_text() == "ok,raised-remote"
    assert (tmp_path / "f1.txt").read_text() == "ok,raised-local"
