#!/usr/bin/env python3
"""The validation pass of the data-parallel loop alone (bench.val_block: attack/uada_ddp.py validate at bs = 8, UADA_ddp.py:233-281) — the form
the rocprofv3 summary profiles/r06_val_kernel_stats.csv is taken with.   python tools/val_bench.py [openvla-7b|tiny] [batches]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from roboticattack_amd import dist as vdist  # noqa: E402

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "openvla-7b"
    batches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = vdist.local_device()
    torch.cuda.set_device(dev)
    model, _ = bench.build_model(kind, dev)
    print(json.dumps(bench._clean(bench.val_block(model, dev, bs=8, batches=batches))))
