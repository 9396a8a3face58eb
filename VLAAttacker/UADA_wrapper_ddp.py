#!/usr/bin/env python3
"""UADA multi-GPU wrapper CLI — drop-in for the reference's VLAAttacker/UADA_wrapper_ddp.py.

    torchrun --nproc_per_node=N --master-addr 127.0.0.1 VLAAttacker/UADA_wrapper_ddp.py --bs 8 ...

Flags :87-106 (no --device, plus --MSE_weights); run dir {cwd}/run/UADA/{exp_id} with exp_id broadcast from rank 0 (:23-35).
Appendix A-D7: the process group is initialised BEFORE the broadcast (the reference forgot to)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from roboticattack_amd import cli  # noqa: E402
from roboticattack_amd import dist as vdist  # noqa: E402
from white_patch.UADA_ddp import OpenVLAAttacker  # noqa: E402


def get_exp_id():
    rank = int(os.environ.get("RANK", 0))
    exp_id = cli.new_exp_id() if rank == 0 else None
    if rank == 0:
        print(f"Generated exp_id on rank 0: {exp_id}")
    return vdist.broadcast_exp_id(exp_id)


def main(args):
    rank, world, local = vdist.env_rank_world()
    device = vdist.local_device()
    if device.type == "cuda":
        torch.cuda.set_device(device)
    vdist.init_process_group(device=device if device.type == "cuda" else None)
    pwd = os.getcwd()
    exp_id = str(get_exp_id())
    vla_path = args.vla_path
    if vla_path is None:
        hub = cli.vla_path_for(args.dataset)
        vla_path = hub if os.path.isdir(hub) else "random:openvla-7b"
    cli.set_seed(42)
    target = "".join(str(i) for i in args.maskidx)
    name = (f"{args.dataset}_modifyLabel_MSEDistance_lr{format(args.lr, '.0e')}_iter{args.iter}_warmup{args.warmup}_target{target}"
            f"_inner_loop{args.innerLoop}_patch_size{args.patch_size}_seed42-{exp_id}")
    cli.maybe_wandb_init(args, name, rank)
    print(f"exp_id:{exp_id}")
    path = f"{pwd}/run/UADA/{exp_id}"
    os.makedirs(path, exist_ok=True)
    instance_params = {
        "vla_path": vla_path, "dataset_name": args.dataset, "save_dir": path, "resize_patch": args.resize_patch,
        "patch_size": args.patch_size, "lr": args.lr, "bs": args.bs, "warmup": args.warmup, "num_iter": args.iter,
        "maskidx": args.maskidx, "innerLoop": args.innerLoop, "geometry": args.geometry,
        "use_wandb": args.wandb_project != "false", "MSE_weights": args.MSE_weights,
    }
    if args.attack != "UADA":  # extension: the same data-parallel loop for UPA / TMA (BASELINE configs 4-5)
        instance_params.update(attack_type=args.attack, alpha=args.alpha, belta=args.belta, target_action=args.targetAction)
    OpenVLAAttacker._attack_entry(rank, instance_params, world)
    print("Attack done!")


def arg_parser(argv=None):
    parser = argparse.ArgumentParser()
    cli.add_common(parser, lr=1e-3, maskidx="0", iters=2000, warmup=20, inner=50, device_default=None, tags=[""])
    parser.add_argument("--MSE_weights", default=5, type=int)
    parser.add_argument("--reverse_direction", type=cli.str2bool, default=True)
    # extensions (not in the reference): run the UPA / TMA objectives in the same data-parallel loop
    parser.add_argument("--attack", default="UADA", choices=["UADA", "UPA", "TMA"])
    parser.add_argument("--alpha", type=float, default=0.8)
    parser.add_argument("--belta", type=float, default=0.2)
    parser.add_argument("--targetAction", default=0, type=float)
    return parser.parse_args(argv)


if __name__ == "__main__":
    args = arg_parser()
    print(f"Paramters:\n maskidx:{args.maskidx}\n lr:{args.lr} \n tags:{args.tags}")
    main(args)
