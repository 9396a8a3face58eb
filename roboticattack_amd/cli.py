"""Shared pieces of the four wrapper CLIs (VLAAttacker/*_wrapper.py): flags, seeding, model/data factories, run dirs.

Flag names, types and defaults are the reference's (UADA_wrapper.py:87-121, UADA_wrapper_ddp.py:87-119,
TMA_wrapper.py:89-124, UPA_wrapper.py:90-127). Added flags (all optional, defaults keep reference behaviour where the
environment allows): --vla_path (local checkpoint dir, or random:openvla-7b / random:tiny / surrogate), --data (synthetic).
"""
from __future__ import annotations

import argparse
import os
import random
import uuid

import numpy as np
import torch


def list_of_ints(arg):
    return list(map(int, arg.split(",")))


def str2bool(value):
    if isinstance(value, bool):
        return value
    if value.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if value.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def set_seed(seed: int):
    """UADA_wrapper.py:15-23."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


DATASET_TO_CKPT = (  # UADA_wrapper.py:29-40
    ("bridge_orig", "openvla/openvla-7b"),
    ("libero_spatial", "openvla/openvla-7b-finetuned-libero-spatial"),
    ("libero_object", "openvla/openvla-7b-finetuned-libero-object"),
    ("libero_goal", "openvla/openvla-7b-finetuned-libero-goal"),
    ("libero_10", "openvla/openvla-7b-finetuned-libero-10"),
)


def vla_path_for(dataset: str) -> str:
    for key, path in DATASET_TO_CKPT:
        if key in dataset:
            return path
    assert False, "Invalid dataset"


def add_common(parser: argparse.ArgumentParser, *, lr, maskidx, iters, warmup, inner, device_default, tags):
    parser.add_argument("--maskidx", default=maskidx, type=list_of_ints)
    parser.add_argument("--lr", default=lr, type=float)
    if device_default is not None:
        parser.add_argument("--device", default=device_default, type=int)
    parser.add_argument("--iter", default=iters, type=int)
    parser.add_argument("--accumulate", default=1, type=int)
    parser.add_argument("--bs", default=8, type=int)
    parser.add_argument("--warmup", default=warmup, type=int)
    parser.add_argument("--tags", nargs="+", default=tags)
    parser.add_argument("--geometry", type=str2bool, nargs="?", default=True, help="add geometry trans to path")
    parser.add_argument("--patch_size", default="3,50,50", type=list_of_ints)
    parser.add_argument("--wandb_project", default="xxx", type=str)
    parser.add_argument("--wandb_entity", default="xxx", type=str)
    parser.add_argument("--innerLoop", default=inner, type=int)
    parser.add_argument("--dataset", default="bridge_orig", type=str)
    parser.add_argument("--resize_patch", type=str2bool, default=False)
    # additions
    parser.add_argument("--vla_path", default=None, type=str,
                        help="local OpenVLA checkpoint dir, or random:openvla-7b | random:tiny | surrogate (no network here)")
    parser.add_argument("--data", default="synthetic", type=str, help="data source; only 'synthetic' exists in this image")


def resolve_device(index: int):
    """`--device N` as the reference uses it (UADA_wrapper.py:52 defaults to GPU 1). On a box with fewer GPUs the run would die with
    'invalid device ordinal' before doing anything; fall back to GPU 0 and say so."""
    import torch

    if not torch.cuda.is_available():
        return torch.device("cpu")
    if index >= torch.cuda.device_count():
        print(f"[vaa] --device {index} does not exist on this host ({torch.cuda.device_count()} GPU(s)) -> using cuda:0")
        index = 0
    return torch.device(f"cuda:{index}")


def resolve_model(args, device):
    """The reference downloads `openvla/...` from the HF hub (UADA_wrapper.py:56-65); offline, a local directory of that
    name is used if present, otherwise the shape-exact random-init model."""
    from .attack.uada_ddp import default_model_factory

    path = args.vla_path
    if path is None:
        hub = vla_path_for(args.dataset)
        path = hub if os.path.isdir(hub) else "random:openvla-7b"
        if path != hub:
            print(f"[vaa] checkpoint {hub!r} not available offline -> random-init OpenVLA-7B-shaped model")
    return default_model_factory(path, device), path


def synthetic_loaders(bs: int, rank: int = 0):
    from .synthetic import SyntheticLoader

    return SyntheticLoader(bs, seed=1234 + 1000003 * rank), SyntheticLoader(bs, seed=99991 + 1000003 * rank)


def maybe_wandb_init(args, name, rank=0):
    if args.wandb_project == "false" or rank != 0:
        return
    try:
        import wandb

        wandb.init(entity=args.wandb_entity, project=args.wandb_project, name=name, tags=args.tags)
        wandb.config = {"iteration": args.iter, "learning_rate": args.lr, "attack_target": args.maskidx, "accumulate_steps": args.accumulate}
    except Exception as e:  # wandb is optional in this image
        print(f"[vaa] wandb unavailable ({e}); continuing with --wandb_project false")
        args.wandb_project = "false"


def new_exp_id() -> str:
    return str(uuid.uuid4())
