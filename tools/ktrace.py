#!/usr/bin/env python3
"""Run a command under `rocprofv3 --kernel-trace` and print avg/min durations of the vaa:: kernels grouped by grid size.
usage (GPU box): python tools/ktrace.py [--all] -- python tools/k1exp.py"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

args = sys.argv[1:]
show_all = False
if args and args[0] == "--all":
    show_all, args = True, args[1:]
if args and args[0] == "--":
    args = args[1:]
d = tempfile.mkdtemp(prefix="ktrace_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
p = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--"] + args, env=env, cwd=os.getcwd(),
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
sys.stdout.write(p.stdout)
fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not fs:
    sys.stderr.write(p.stderr[-2000:])
    sys.exit(1)
acc = collections.OrderedDict()
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if not show_all and "vaa::" not in n:
        continue
    key = (n.split("(")[0][-48:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v2 = sorted(v)
    print(f"{k[0]:50s} grid={k[1]:>9s} wg={k[2]:>5s} n={len(v):4d} avg={sum(v)/len(v):9.2f}us med={v2[len(v2)//2]:9.2f} min={v2[0]:9.2f}")
shutil.rmtree(d, ignore_errors=True)
