#!/usr/bin/env python3
"""HBM-side traffic of the hand-written kernels from rocprofv3 PMC counters (run on the MI355X box).

Recipe of /opt/skills/guides/MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE passes (they do not fit one pass: 3 + 2 of 4 TCC slots) with nothing but --pmc on the command line; both are
in KiB at the L2's memory-side (fabric) interface; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x and the
other widths / WRITE_SIZE are uncalibrated, so every figure is corrected by factors measured IN THE SAME RUN on a
calibration copy kernel of known size (512 MiB device-to-device `copy_`: 512 MiB read + 512 MiB written, larger than the
256 MiB Infinity Cache).

    python tools/pmc_traffic.py            # writes gpurun_out/traffic.json (copy it to profiles/traffic_rNN.json)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB_BYTES = 512 * 1024 * 1024


def run_pass(counter):
    d = tempfile.mkdtemp(prefix=f"pmc_{counter}_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "tools", "kbench.py"), "--iters", "3", "--calib"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        sys.stderr.write(p.stderr[-3000:])
        raise SystemExit(f"no counter_collection.csv for {counter}")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        key = name.split("(")[0][-110:] + "|" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        acc[key].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    return acc


def main():
    out = {}
    raw = {c: run_pass(c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    # calibration: the largest elementwise copy kernel of the run
    def calib(counter):
        best = None
        for k, v in raw[counter].items():
            if "vaa::" in k:
                continue
            m = max(v)
            if best is None or m > best[1]:
                best = (k, m)
        return best

    cf, cw = calib("FETCH_SIZE"), calib("WRITE_SIZE")
    f_fac = CALIB_BYTES / (cf[1] * 1024.0)
    w_fac = CALIB_BYTES / (cw[1] * 1024.0)
    out["_calibration"] = {"copy_bytes_each_way": CALIB_BYTES, "fetch_kernel": cf[0], "FETCH_SIZE_KiB": cf[1], "fetch_factor": f_fac,
                           "write_kernel": cw[0], "WRITE_SIZE_KiB": cw[1], "write_factor": w_fac}
    keys = sorted(set(k for k in raw["FETCH_SIZE"] if "vaa::" in k) | set(k for k in raw["WRITE_SIZE"] if "vaa::" in k))
    for k in keys:
        f = raw["FETCH_SIZE"].get(k, [])
        w = raw["WRITE_SIZE"].get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out[k] = {"launches": len(f), "FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm, "read_bytes": fm * 1024 * f_fac,
                  "write_bytes": wm * 1024 * w_fac, "hbm_bytes_per_launch": fm * 1024 * f_fac + wm * 1024 * w_fac}
    # per hot-path op (what bench.py's event brackets cover): sum over the op's kernels
    ops = {"K1_patch_apply_fwd": ("patch_apply_fwd_kernel",),
           "K1t_patch_apply_fwd_tiles": ("patch_apply_tiles_kernel",),                                            # K1 in tile-major form (what the attack step runs)
           "K2_patch_grad_gather": ("patch_grad_scatter_kernel<3, false, false", "patch_grad_reduce_kernel|30208"),
           "K2_patch_embed_grad_gather": ("embed_dgrad_tiles", "patch_grad_scatter_kernel<3, true, false", "patch_grad_reduce_kernel|30208"),
           "K2et_deferred_reduce": ("embed_dgrad_tiles", "patch_grad_scatter_kernel<3, true, false"),             # tile GEMM + gather; the final sum is the epilogue's
           "K3_loss_rows_fwd_bwd": ("rows_stats_kernel<unsigned short, 256, false>", "rows_finish_kernel<unsigned short, 256>|256"),        # UADA_DDP: gradient slice, one finishing workgroup
           "K3s_loss_rows_stats": ("rows_stats_kernel<unsigned short, 256, false>",),
           "K3_full_one_launch_optin": ("rows_stats_kernel<unsigned short, 256, true>",),                          # UADA: full-row gradient in ONE launch (VAA_K3_ONE_PASS=1, rows read once)
           "K3_full_rows_fwd_bwd": ("rows_stats_kernel<unsigned short, 256, false>", "rows_finish_kernel<unsigned short, 256>|131072"),  # the default since round 4: two launches
           "K3h_head_loss_rows_stats": ("head_stats_kernel", "head_finish_kernel"),                                   # LM head fused with K3's statistics (weight stream)
           "K3sl_head_slice_fwd_bwd": ("head_slice_kernel",),                                                        # K3s: slice-only head + statistics + gradient + head backward
           "EPI_step_epilogue": ("step_epilogue_kernel",),
           "K4_patch_update": ("patch_update_kernel",)}
    # a kernel that belongs to two ops (reduce: K2 and K2'; stats: both K3 modes) ran once per op call, so its per-launch mean is counted once in each
    out["ops"] = {}
    for op, kns in ops.items():
        tot = sum(v["hbm_bytes_per_launch"] for k, v in out.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v and any(kn in k for kn in kns))
        out["ops"][op] = {"hbm_bytes_per_launch": tot, "kernels": list(kns)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
