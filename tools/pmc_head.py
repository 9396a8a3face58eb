#!/usr/bin/env python3
"""SQ counters of the fused LM-head kernel (GPU box). usage: pmc_head.py run R | pmc R CTR..."""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def run(R):
    import torch
    from roboticattack_amd import ops, synthetic
    from roboticattack_amd.labels import mask_labels
    dev = torch.device("cuda:0")
    D, V = 4096, 32064
    g = torch.Generator(device=dev).manual_seed(0)
    W = (torch.randn(V, D, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    _, labels, _ = synthetic.synth_text_batch(4242, R // 2)
    labels = mask_labels(labels, [0]).to(dev)
    rm = ops.LossRowMap(labels)
    h = torch.randn(R, D, device=dev, generator=g).to(torch.bfloat16)
    gs = torch.empty((R, 256), dtype=torch.bfloat16, device=dev)
    for _ in range(6):
        ops.head_loss_rows_stats(h, W, rm, ops.LOSS_UADA_DDP, 5.0, grad=gs)
    torch.cuda.synchronize()
if sys.argv[1] == "run":
    run(int(sys.argv[2]))
else:
    d = tempfile.mkdtemp(prefix="head_", dir="/tmp")
    p = subprocess.run(["rocprofv3", "--pmc", *sys.argv[3:], "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "run", sys.argv[2]],
                       env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        print(p.stderr[-1500:]); sys.exit(1)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "head_stats" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
    shutil.rmtree(d, ignore_errors=True)
