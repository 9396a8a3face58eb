#!/usr/bin/env python3
"""UADA single-GPU wrapper CLI — drop-in for the reference's VLAAttacker/UADA_wrapper.py (flags :87-108, run dir :50)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from roboticattack_amd import cli  # noqa: E402
from white_patch.UADA import OpenVLAAttacker  # noqa: E402


def main(args):
    pwd = os.getcwd()
    exp_id = cli.new_exp_id()
    cli.vla_path_for(args.dataset)  # "Invalid dataset" assertion of the reference
    cli.set_seed(42)
    target = "".join(str(i) for i in args.maskidx)
    name = (f"{args.dataset}_modifyLabel_MSEDistance_lr{format(args.lr, '.0e')}_iter{args.iter}_warmup{args.warmup}_target{target}"
            f"_inner_loop{args.innerLoop}_patch_size{args.patch_size}_seed42-{exp_id}")
    cli.maybe_wandb_init(args, name)
    print(f"exp_id:{exp_id}")
    path = f"{pwd}/run/UADA/{exp_id}"
    device = cli.resolve_device(args.device)
    vla, _ = cli.resolve_model(args, device)
    os.makedirs(path, exist_ok=True)
    train_dataloader, val_dataloader = cli.synthetic_loaders(args.bs)
    attacker = OpenVLAAttacker(vla, None, path, optimizer="adamW", resize_patch=args.resize_patch)
    attacker.patchattack_unconstrained(train_dataloader, val_dataloader, num_iter=args.iter, target_action=np.zeros(7),
                                       patch_size=args.patch_size, lr=args.lr, accumulate_steps=args.accumulate, maskidx=args.maskidx,
                                       warmup=args.warmup, filterGripTrainTo1=args.filterGripTrainTo1, geometry=args.geometry,
                                       innerLoop=args.innerLoop, args=args)
    print("Attack done!")


def arg_parser(argv=None):
    parser = argparse.ArgumentParser()
    cli.add_common(parser, lr=1e-3, maskidx="0", iters=2000, warmup=20, inner=50, device_default=1, tags=[""])
    parser.add_argument("--filterGripTrainTo1", type=cli.str2bool, nargs="?", default=False,
                        help="Remove the gripper 0 traning samples during the attack of target at grip to 0")
    parser.add_argument("--reverse_direction", type=cli.str2bool, default=True)
    return parser.parse_args(argv)


if __name__ == "__main__":
    args = arg_parser()
    print(f"Paramters:\n maskidx:{args.maskidx}\n lr:{args.lr} \n device:{args.device} \ntags:{args.tags}")
    main(args)
