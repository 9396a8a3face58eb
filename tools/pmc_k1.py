#!/usr/bin/env python3
"""SQ / TA / TCC counters of the K1 kernel (one rocprofv3 --pmc pass per invocation; GPU box).
usage: pmc_k1.py run B | pmc_k1.py pmc B CTR [CTR...]"""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def run(B):
    import torch
    from roboticattack_amd import benchmarks, ops, synthetic
    dev = torch.device("cuda:0")
    img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    for _ in range(6):
        ops.patch_apply_fwd(img, patch, xy, th, True)
    torch.cuda.synchronize()
if sys.argv[1] == "run":
    run(int(sys.argv[2]))
else:
    B = sys.argv[2]
    d = tempfile.mkdtemp(prefix="pmck1_", dir="/tmp")
    p = subprocess.run(["rocprofv3", "--pmc", *sys.argv[3:], "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "run", B],
                       env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(p.stderr[-1500:]); sys.exit(1)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "patch_apply_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("B", B, {k: round(sum(v[1:]) / max(1, len(v) - 1)) for k, v in acc.items()})
    shutil.rmtree(d, ignore_errors=True)
