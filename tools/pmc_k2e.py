#!/usr/bin/env python3
"""Per-kernel time / SQ counters of the K2' kernels (GPU box). usage: pmc_k2e.py run | trace | pmc CTR..."""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def run():
    import torch
    from roboticattack_amd import benchmarks, ops, synthetic
    dev = torch.device("cuda:0")
    B, D0, D1 = 64, 1024, 1152
    img = torch.from_numpy(synthetic.synth_images(1, B, "noise")).to(dev)
    patch = torch.rand(3, 50, 50, device=dev)
    xy_n, th_n = benchmarks.random_params(B, 50, 50, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    _, keep = ops.patch_apply_fwd(img, patch, xy, th, True, want_keep=True)
    dy0 = (torch.randn(B, 256, D0, device=dev) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, D1, device=dev) * 0.1).to(torch.bfloat16)
    wt0 = ops.pack_embed_weights((torch.randn(588, D0, device=dev) * 0.05).to(torch.bfloat16))
    wt1 = ops.pack_embed_weights((torch.randn(588, D1, device=dev) * 0.05).to(torch.bfloat16))
    for _ in range(6):
        ops.patch_embed_grad_gather(dy0, dy1, wt0, wt1, patch, xy, th, keep, True)
    torch.cuda.synchronize()
def prof(args):
    d = tempfile.mkdtemp(prefix="k2e_", dir="/tmp")
    p = subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "run"],
                       env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return d, p
if sys.argv[1] == "run":
    run()
elif sys.argv[1] == "trace":
    d, p = prof(["--kernel-trace", "--stats"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "vaa::" in r["Name"]:
                print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
    shutil.rmtree(d, ignore_errors=True)
else:
    d, p = prof(["--pmc", *sys.argv[2:]])
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(p.stderr[-1500:]); sys.exit(1)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "embed" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
    shutil.rmtree(d, ignore_errors=True)
