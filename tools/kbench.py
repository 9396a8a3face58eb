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
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    from roboticattack_amd import benchmarks, ops

    ops.device_check()
    ph, pw = [int(v) for v in args.patch.split(",")]
    res = {"suite": benchmarks.kernel_suite(args.bs, ph, pw, iters=args.iters)}
    if args.sweep:
        res["k2_sweep"] = benchmarks.k2_sweep(ph=ph, pw=pw, iters=max(5, args.iters // 2))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
