#!/usr/bin/env python3
"""SQ / memory counters of the model-side attention kernels (GPU box): one `rocprofv3 --pmc` pass over tools/attn_bench.py.
    python tools/pmc_attn.py <llm|dino|siglip> CTR [CTR ...]        (counters only: no trace options on the same command line)"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shape, ctrs = sys.argv[1], sys.argv[2:]
d = tempfile.mkdtemp(prefix="attn_", dir="/tmp")
p = subprocess.run(["rocprofv3", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "attn_bench.py"),
                    "--shape", shape, "--iters", "2"], env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not fs:
    print(p.stderr[-1500:])
    sys.exit(1)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if "attn_" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, flush=True)
shutil.rmtree(d, ignore_errors=True)
