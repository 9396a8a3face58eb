#!/usr/bin/env python3
"""Kernel-only microbench of the hand-written HIP kernels (no model). Used for rocprofv3 kernel-trace / PMC passes:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k -- python tools/kbench.py --iters 20
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python tools/kbench.py --iters 5
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--patch", type=str, default="50,50")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--calib", action="store_true", help="also run a 512 MiB device copy (PMC calibration kernel)")
    args = ap.parse_args()
    from roboticattack_amd import benchmarks, ops

    ops.device_check()
    ph, pw = [int(v) for v in args.patch.split(",")]
    if args.calib:
        import torch

        src = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        del src, dst
    res = {"suite": benchmarks.kernel_suite(args.bs, ph, pw, iters=args.iters)}
    if args.sweep:
        res["k2_sweep"] = benchmarks.k2_sweep(ph=ph, pw=pw, iters=max(5, args.iters // 2))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
