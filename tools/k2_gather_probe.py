#!/usr/bin/env python3
"""The gather ("one lane per patch texel") formulation of K2 against the product's scatter: parity and time (GPU box).

    python tools/k2_gather_probe.py            # builds tools/probe/k2_gather_probe.hip -> tools/probe/libk2gather.so if missing
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roboticattack_amd import benchmarks, ops, synthetic  # noqa: E402

SO = os.path.join(ROOT, "tools", "probe", "libk2gather.so")


def build():
    src = os.path.join(ROOT, "tools", "probe", "k2_gather_probe.hip")
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(src):
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", src, "-o", SO])


def main():
    build()
    if not torch.cuda.is_available():
        print("built", SO)
        return
    ops.device_check()
    L = C.CDLL(SO)
    L.k2_gather_probe.restype = C.c_int
    L.k2_gather_probe.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.POINTER(C.c_float)] + [C.c_void_p] * 3
    dev = torch.device("cuda:0")
    std6 = (C.c_float * 6)(0.228515625, 0.2236328125, 0.224609375, 0.5, 0.5, 0.5)
    ph = pw = 50
    for B in (8, 64, 256, 1024, 4096):
        patch = torch.rand(3, ph, pw, device=dev)
        xy_n, th_n = benchmarks.random_params(B, ph, pw, 42)
        xy_n = np.clip(xy_n, 1, 224 - 50 - 1).astype(np.int32)  # interior placements: the probe has no edge-ray pass
        xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
        img = torch.zeros((B, 224, 224, 3), dtype=torch.uint8, device=dev)
        _, keep = ops.patch_apply_fwd(img, patch, xy, th, True)
        g = synthetic.synth_upstream_grad(7, 64).to(dev).repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
        ref = ops.patch_grad_gather(g, patch, xy, th, keep, True)
        best = None
        for S in sorted({min(B, 16), min(B, 64), min(B, 256)}):
            partial = torch.empty(S * 3 * ph * pw, dtype=torch.float32, device=dev)
            out = torch.empty(3, ph, pw, dtype=torch.float32, device=dev)

            def run():
                rc = L.k2_gather_probe(g.data_ptr(), xy.data_ptr(), th.data_ptr(), keep.data_ptr(), B, ph, pw, S, std6, partial.data_ptr(),
                                       out.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc

            run()
            torch.cuda.synchronize()
            err = float((out - ref).abs().max() / ref.abs().max())
            t = benchmarks._time(run, 20)[0] * 1e6
            if best is None or t < best[1]:
                best = (S, t, err)
        t_ref = benchmarks._time(lambda: ops.patch_grad_gather(g, patch, xy, th, keep, True), 20)[0] * 1e6
        print(f"B={B:5d}  gather (texel lanes, S={best[0]:3d} image slots): {best[1]:7.1f} us   product scatter: {t_ref:7.1f} us   max |diff| / max |g| = {best[2]:.2e}")


if __name__ == "__main__":
    main()
