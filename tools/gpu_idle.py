#!/usr/bin/env python3
"""GPU idle time inside the attack step, from a rocprofv3 --kernel-trace CSV: the union of the busy intervals of all dispatches between two
K1 launches (one step), against the step's wall time.  python tools/gpu_idle.py <kernel_trace.csv> [skip_steps]"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
k1 = [i for i, e in enumerate(ev) if "patch_apply_tiles_kernel" in e[2]]
out = []
for a, b in zip(k1[skip:-1], k1[skip + 1:]):
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += min(cur_e, t1) - cur_s
    out.append(((t1 - t0) / 1e6, busy / 1e6, len(seg), sum(g for g in gaps if g > 20000) / 1e6, sum(1 for g in gaps if g > 20000)))
a = np.array(out)
print(f"steps {len(out)}: wall {a[:,0].mean():.2f} ms, busy {a[:,1].mean():.2f} ms ({100*a[:,1].mean()/a[:,0].mean():.1f} %), {a[:,2].mean():.0f} dispatches, "
      f"gaps > 20 us: {a[:,4].mean():.0f} per step = {a[:,3].mean():.2f} ms")
