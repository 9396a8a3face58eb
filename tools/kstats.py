#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace --stats` CSV of a bench.py run: GPU kernel time per step, the hand-written kernels of the hot
path (calls, average us, us per step, algorithmic share) and the top framework kernels.

    python tools/kstats.py <kernel_stats.csv> <steps incl. warm-up and profiled pass> [--top N]
"""
import csv
import sys

HAND = ("patch_apply", "embed_dgrad", "patch_grad_", "head_slice", "head_stats_kernel", "head_finish_kernel", "rows_stats_kernel", "rows_finish_kernel", "loss_stats_kernel",
        "loss_grad_kernel", "step_epilogue_kernel", "patch_update_kernel", "patch_resize", "loss_rowmap_kernel")


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 8
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    hand = [r for r in rows if any(h in r["Name"] for h in HAND)]
    hand_tot = sum(float(r["TotalDurationNs"]) for r in hand)
    print(f"GPU kernel time per step: {tot / steps / 1e6:.3f} ms over {steps} steps; hand-written path: {hand_tot / steps / 1e3:.1f} us per step ({100 * hand_tot / tot:.3f} %)")
    for r in sorted(hand, key=lambda r: -float(r["TotalDurationNs"])):
        print(f'  {int(r["Calls"]):>5} calls  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {float(r["TotalDurationNs"]) / steps / 1e3:8.2f} us/step  {r["Name"][:110]}')
    print(f"top {top} other kernels:")
    for r in sorted((r for r in rows if r not in hand), key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        print(f'  {int(r["Calls"]):>5} calls  avg {float(r["AverageNs"]) / 1e3:8.2f} us  {float(r["TotalDurationNs"]) / steps / 1e6:8.3f} ms/step  {r["Name"][:100]}')


if __name__ == "__main__":
    main()
