#!/usr/bin/env python3
"""SQ counters of the K2 scatter kernel at B=4096 (one rocprofv3 --pmc pass; GPU box)."""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctrs = sys.argv[1:] or ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
d = tempfile.mkdtemp(prefix="pmck2_", dir="/tmp")
p = subprocess.run(["rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "k2exp.py")],
                   env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print(p.stderr[-2000:]); sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "scatter" in r["Kernel_Name"]:
        acc[r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, cs in acc.items():
    print("grid", g, {k: round(sum(v) / len(v)) for k, v in cs.items()})
shutil.rmtree(d, ignore_errors=True)
