#!/usr/bin/env python3
"""Per-kernel time and SQ counters of the attention kernels at one shape (GPU box).
usage: attn_prof.py run [llm|dino|siglip]      -> executes the three kernels a few times (the profiled workload)
       attn_prof.py trace [shape]              -> rocprofv3 --kernel-trace --stats summary of `run`
       attn_prof.py pmc [shape] CTR [CTR...]   -> one rocprofv3 --pmc pass, averaged per kernel"""
import collections, csv, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {"llm": (64, 32, 300, 128, True, False), "dino": (64, 16, 261, 64, False, True), "siglip": (64, 16, 256, 72, False, True)}

def run(shape, iters=5):
    import torch
    from roboticattack_amd import model_ops
    B, H, T, hd, causal, packed = SHAPES[shape]
    if packed:
        qkv = torch.randn(B, T, 3, H, hd, device="cuda").to(torch.bfloat16)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = [torch.randn(B, T, H, hd, device="cuda").to(torch.bfloat16) for _ in range(3)]
    go = torch.randn(B, T, H, hd, device="cuda").to(torch.bfloat16)
    for _ in range(iters):
        o, lse = model_ops.attention_fwd(q, k, v, causal, hd ** -0.5)
        model_ops.attention_bwd(q, k, v, o, lse, go, causal, hd ** -0.5, packed_grad=packed)
    torch.cuda.synchronize()

def prof(args, shape):
    d = tempfile.mkdtemp(prefix="attnprof_", dir="/tmp")
    p = subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "run", shape],
                       env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return d, p

if __name__ == "__main__":
    mode = sys.argv[1]
    shape = sys.argv[2] if len(sys.argv) > 2 else "llm"
    if mode == "run":
        run(shape)
    elif mode == "trace":
        d, p = prof(["--kernel-trace", "--stats"], shape)
        fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
        if not fs:
            print(p.stderr[-2000:]); sys.exit(1)
        for r in csv.DictReader(open(fs[0])):
            if "attn" in r["Name"]:
                print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
        shutil.rmtree(d, ignore_errors=True)
    else:
        d, p = prof(["--pmc", *sys.argv[3:]], shape)
        fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            print(p.stderr[-2000:]); sys.exit(1)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[0])):
            if "attn" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("<")[0].split("(")[0][-24:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kname, cs in acc.items():
            print(kname, {k: round(sum(v) / len(v)) for k, v in cs.items()})
        shutil.rmtree(d, ignore_errors=True)
