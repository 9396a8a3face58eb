"""Kernel-level measurement helpers shared by bench.py and tools/kbench.py (HIP events on the launching stream).

Algorithmic bytes per launch (restated in DESIGN.md):
  K1 fwd   : B*(150,528 [u8 frame] + 602,112 [bf16 6-ch output]) + 12*ph*pw [patch]
  K2 gather: strict  = B*3*ph*pw*(2*2 [bf16 upstream grad, channels c and c+3, ~ph*pw kept pixels/|det|~1]) + 12*ph*pw
             i.e. ~ B*12*ph*pw + 12*ph*pw  (B*30,000 + 30,000 at 50x50);  full-frame figure B*602,112 reported separately
  K3 loss  : 2 * R' * V * e   (one read + one write of the labelled rows; e = bytes per logit); "K3_slice" (UADA_DDP / UPA, gradient
             confined to the 256 action columns): R' * V * e read + R' * 256 * e written
  K4 update: 7*4*n  (patch, g, m, v read; patch, m, v written)
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops, synthetic
from .labels import mask_labels

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling


def algo_bytes(kernel: str, B: int, ph: int = 50, pw: int = 50, rows: int = 0, V: int = 32064, esize: int = 2, embed_width: int = 1024 + 1152) -> float:
    n = 3 * ph * pw
    if kernel == "K1":
        return B * (150528 + 602112) + 4 * n
    if kernel == "K2":
        return B * 4 * n + 4 * n
    if kernel == "K2_fullframe":
        return B * 602112 + 4 * n
    if kernel == "K2e":  # ~21 flagged tiles per image at 50x50 (20.5 measured; rounds 1-3 used K1's conservative footprint count, 36): their dY rows of
        # both towers (bf16) + the two transposed weights once + gpatch
        tiles = max(1.0, (np.sqrt(float(ph) * pw) / 14.0 + 1.0) ** 2)
        return int(B * tiles * embed_width * 2) + 588 * embed_width * 2 + 4 * n
    if kernel == "K3":
        return 2.0 * rows * V * esize
    if kernel == "K3_slice":
        return rows * V * esize + rows * 256 * esize
    if kernel == "K4":
        return 7 * 4 * n
    raise KeyError(kernel)


def random_params(B, ph, pw, seed=0):
    """Host draws in the reference's order (appply_random_transform.py:120-128) -> device tensors."""
    import random

    from .transform import RandomPatchTransform

    st_r, st_n = random.getstate(), np.random.get_state()
    random.seed(seed)
    np.random.seed(seed)
    t = RandomPatchTransform("cpu")
    xy, th = t._draw(B, ph, pw, True)
    random.setstate(st_r)
    np.random.set_state(st_n)
    return xy, th


def _time_in_stream(fn, iters, warmup=3, reps=5):
    """Mean GPU time of one `fn()` call launched back to back in the current stream: `iters` calls between two HIP events, `reps` times. For
    ops that cannot be captured into a graph (K3's one-pass form: its hand-over generation is a launch argument). The host enqueues faster
    than these kernels run once a few are queued, so the bracket is GPU time."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3 / iters)
    t = np.array(ts)
    return float(t.mean()), float(np.median(t)), float(t.min())


def _time(fn, iters, warmup=3, per_graph=10):
    """Mean GPU time of one `fn()` call. The launch sequence is captured `per_graph` times into a hipGraph and the
    graph is replayed `iters` times between two HIP events on the replay stream, so host launch overhead (python +
    ctypes, ~10 us per call) cannot pollute kernels that only run for a few microseconds. The figure agrees with
    rocprofv3's per-kernel durations plus the dependent-launch gaps of multi-kernel ops. Falls back to per-call event
    pairs if capture is not possible."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            with torch.cuda.graph(g, stream=side):
                for _ in range(per_graph):
                    fn()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(max(3, iters // per_graph)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e-3 / per_graph)
        t = np.array(ts)
        return float(t.mean()), float(np.median(t)), float(t.min())
    except Exception:  # pragma: no cover - capture unsupported
        torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    t = np.array([s.elapsed_time(e) for s, e in evs]) * 1e-3
    return float(t.mean()), float(np.median(t)), float(t.min())


def kernel_suite(B=64, ph=50, pw=50, iters=50, device="cuda:0", maskidx=(0,), logits_dtype=torch.bfloat16):
    """Times K1..K4 standalone on BASELINE-shaped synthetic inputs already resident in HBM. Returns dict per kernel."""
    dev = torch.device(device)
    img = torch.from_numpy(synthetic.synth_images(1234, B, "noise")).to(dev)
    patch = torch.rand(3, ph, pw, device=dev)
    xy_n, th_n = random_params(B, ph, pw, 42)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    out, keep = ops.patch_apply_fwd(img, patch, xy, th, True)
    g = synthetic.synth_upstream_grad(7, min(B, 64)).to(dev)
    if B > g.shape[0]:
        g = g.repeat((B + g.shape[0] - 1) // g.shape[0], 1, 1, 1)[:B].contiguous()
    _, labels, _ = synthetic.synth_text_batch(4242, B)
    labels = mask_labels(labels, list(maskidx)).to(dev)
    R = int((labels[:, 1:] != -100).sum())
    logits = (torch.randn(R, 32064, device=dev) * 2).to(logits_dtype)
    glog = torch.empty_like(logits)
    m, v = torch.zeros_like(patch), torch.zeros_like(patch)
    gp = torch.randn_like(patch) * 1e-3
    res = {}

    def rec(name, key, fn, nbytes, in_stream=False, **extra):
        mean, med, mn = _time_in_stream(fn, max(iters, 100)) if in_stream else _time(fn, iters)
        res[name] = dict(mean_us=mean * 1e6, median_us=med * 1e6, min_us=mn * 1e6, algo_bytes=nbytes,
                         achieved_GBs=nbytes / mean / 1e9, frac_of_8TBs=nbytes / mean / 1e9 / HBM_PEAK_GBS, **extra)

    rec("K1_patch_apply_fwd", "K1", lambda: ops.patch_apply_fwd(img, patch, xy, th, True), algo_bytes("K1", B, ph, pw), B=B)
    rec("K2_patch_grad_gather", "K2", lambda: ops.patch_grad_gather(g, patch, xy, th, keep, True), algo_bytes("K2", B, ph, pw), B=B,
        fullframe_bytes=algo_bytes("K2_fullframe", B, ph, pw))
    dy0 = (torch.randn(B, 256, 1024, device=dev) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, 1152, device=dev) * 0.1).to(torch.bfloat16)
    wt0 = ops.pack_embed_weights((torch.randn(588, 1024, device=dev) * 0.05).to(torch.bfloat16))
    wt1 = ops.pack_embed_weights((torch.randn(588, 1152, device=dev) * 0.05).to(torch.bfloat16))
    rec("K2e_patch_embed_grad_gather", "K2e", lambda: ops.patch_embed_grad_gather(dy0, dy1, wt0, wt1, patch, xy, th, keep, True),
        algo_bytes("K2e", B, ph, pw), B=B, note="SURVEY 8f-3: patch-embed backward on the kept tiles (MFMA) + gather; replaces 2 dgrad GEMMs + fold + K2")
    # the tile-major forms the attack step runs with a model that exposes its patch-embed weights
    rec("K1t_patch_apply_fwd_tiles", "K1", lambda: ops.patch_apply_fwd_tiles(img, patch, xy, th, True), algo_bytes("K1", B, ph, pw), B=B,
        note="K1 writing the two patch-embed GEMM operands [B,256,588] + tile-major keep words + tile flags (no im2col copies)")
    _, _, keep_t, tflags = ops.patch_apply_fwd_tiles(img, patch, xy, th, True)
    rec("K2et_patch_embed_grad_gather_tiles", "K2e", lambda: ops.patch_embed_grad_gather_tiles(dy0, dy1, wt0, wt1, patch, xy, th, keep_t, tflags, True),
        algo_bytes("K2e", B, ph, pw), B=B, note="K2' fed by K1's tile flags / keep words")
    rec("K2et_deferred_reduce", "K2e", lambda: ops.patch_embed_grad_gather_tiles(dy0, dy1, wt0, wt1, patch, xy, th, keep_t, tflags, True, defer_reduce=True),
        algo_bytes("K2e", B, ph, pw), B=B, note="tile GEMM + scatter only: the final sum is the step epilogue's")
    rowmap = ops.LossRowMap(labels)
    gslice = torch.empty((R, 256), dtype=logits_dtype, device=dev)
    rec("K3_loss_rows_fwd_bwd", "K3_slice",
        lambda: ops.loss_rows_fwd_bwd(logits, rowmap, ops.LOSS_UADA_DDP, w=5.0, grad_kind=ops.GRAD_SLICE, grad=gslice),
        algo_bytes("K3_slice", B, rows=R, esize=logits.element_size()), rows=R,
        note="the bench step's mode (UADA_DDP): row map prebuilt, gradient = the 256 action columns; 16.4 MB figure of SURVEY 8d = full-row storage, see K3_full")
    def k3_full(one_pass: str):
        import os

        def run():
            old = os.environ.get("VAA_K3_ONE_PASS")
            os.environ["VAA_K3_ONE_PASS"] = one_pass
            try:
                ops.loss_rows_fwd_bwd(logits, rowmap, ops.LOSS_UADA, w=5.0, grad_kind=ops.GRAD_FULL, grad=glog)
            finally:
                if old is None:
                    del os.environ["VAA_K3_ONE_PASS"]
                else:
                    os.environ["VAA_K3_ONE_PASS"] = old
        return run

    rec("K3_full_rows_fwd_bwd", "K3", k3_full("0"), algo_bytes("K3", B, rows=R, esize=logits.element_size()), in_stream=True, rows=R,
        note="UADA (1/CE term): full-row gradient, SURVEY 8d's 2*R'*V*e; the DEFAULT form since round 4: statistics launch + finishing launch that reads "
             "every row again; timed as back-to-back calls in a stream")
    rec("K3_full_one_launch_optin", "K3", k3_full("1"), algo_bytes("K3", B, rows=R, esize=logits.element_size()), in_stream=True, rows=R,
        note="the same with VAA_K3_ONE_PASS=1 (opt-in: statistics, grid-wide hand-over, gradient from the registers — every row read once; relies on the "
             "whole grid being resident, fails loudly through vaa_async_error when the hand-over times out)")
    parts = ops.patch_embed_grad_gather_tiles(dy0, dy1, wt0, wt1, patch, xy, th, keep_t, tflags, True, defer_reduce=True)
    msg, scal8 = torch.zeros(3 * ph * pw + 4, device=dev), torch.zeros(8, device=dev)
    ws3 = ops.loss_rows_stats(logits, rowmap, ops.LOSS_UADA_DDP, w=5.0, grad=gslice)
    rec("K3s_loss_rows_stats", "K3_slice", lambda: ops.loss_rows_stats(logits, rowmap, ops.LOSS_UADA_DDP, w=5.0, grad=gslice),
        algo_bytes("K3_slice", B, rows=R, esize=logits.element_size()), rows=R, note="K3 statistics + gradient slice (the fold is the epilogue's)")
    rec("EPI_step_epilogue", "K2", lambda: ops.step_epilogue(parts, msg, scal8, rowmap=rowmap, R=R, V=32064, mode=ops.LOSS_UADA_DDP, w=5.0, loss_ws=ws3),
        parts.numel() * 4 + 4 * 3 * ph * pw, note="K2's final fixed-order sum + K3's fold + the DDP message in one launch")
    rec("K4_patch_update", "K4", lambda: ops.patch_update(patch, gp, m, v, ops.OPT_ADAMW_HF, 1e-3, 1), algo_bytes("K4", B, ph, pw))
    # SURVEY 8f-2 as written: LM head (OpenVLA-7B: [32064, 4096] bf16 = 263 MB) fused with K3's statistics — what the attack step runs up to 64
    # labelled rows (the per-rank shapes of the multi-GPU configs); a weight stream: algorithmic bytes = the head weight + the hidden rows, once
    Dh = 4096
    if ops.head_loss_rows_applies(R, Dh, 32064):
        gh = torch.Generator(device=dev).manual_seed(3)
        w_head = (torch.randn(32064, Dh, device=dev, generator=gh) * 0.02).to(torch.bfloat16)
        hid = torch.randn(R, Dh, device=dev, generator=gh).to(torch.bfloat16)
        nb = 2 * 32064 * Dh + 2 * R * Dh
        rec("K3h_head_loss_rows_stats", "K3h", lambda: ops.head_loss_rows_stats(hid, w_head, rowmap, ops.LOSS_UADA_DDP, 5.0, grad=gslice), nb, in_stream=True,
            rows=R, note="LM head + K3 statistics + gradient slice in two launches (head_stats_kernel + head_finish_kernel), back-to-back calls in a stream: "
                         "the 263 MB of weights partly stay in the 256 MB Infinity Cache between calls — the in-step figure is the bench line's `roofline`")
        rec("K3h_gemm_path_for_comparison", "K3h", lambda: ops.loss_rows_stats(torch.nn.functional.linear(hid, w_head), rowmap, ops.LOSS_UADA_DDP, 5.0, grad=gslice),
            nb, in_stream=True, rows=R, note="the same through the hipBLASLt LM-head GEMM + vaa_loss_rows_stats (what the step runs with VAA_FUSED_HEAD=0)")
        if ops.head_slice_applies(R, Dh, 32064):
            # K3s: the slice-only head — logits of the 256 action columns + statistics + gradient + head backward in ONE launch: hidden rows in, d hidden
            # out, the 2.1 MB weight slice and its transposed copy
            nbs = 2 * (2 * R * Dh) + 2 * (2 * 256 * Dh)
            rec("K3s_head_slice_fwd_bwd", "K3s", lambda: ops.head_slice_fwd_bwd(hid, w_head, rowmap, ops.LOSS_UADA_DDP, 5.0, want_scalars=False), nbs, in_stream=True,
                rows=R, note="what every inner step of the slice modes runs (K3h only behind it on the steps whose CE is read); back-to-back calls in a stream")
        del w_head, hid
    return res


def k2_sweep(batches=(64, 256, 1024, 4096), ph=50, pw=50, iters=20, device="cuda:0"):
    """The north-star roofline target of K2 is evaluated on a batch sweep (one launch = B images; §8d)."""
    dev = torch.device(device)
    out = []
    patch = torch.rand(3, ph, pw, device=dev)
    g64 = synthetic.synth_upstream_grad(7, 64).to(dev)
    for B in batches:
        xy_n, th_n = random_params(B, ph, pw, 42)
        xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
        img = torch.zeros((B, 224, 224, 3), dtype=torch.uint8, device=dev)
        _, keep = ops.patch_apply_fwd(img, patch, xy, th, True)
        del img
        g = g64.repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
        mean, med, mn = _time(lambda: ops.patch_grad_gather(g, patch, xy, th, keep, True), iters)
        nb = algo_bytes("K2", B, ph, pw)
        out.append(dict(B=B, mean_us=mean * 1e6, min_us=mn * 1e6, algo_bytes=nb, achieved_GBs=nb / mean / 1e9,
                        frac_of_8TBs=nb / mean / 1e9 / HBM_PEAK_GBS, fullframe_GBs=algo_bytes("K2_fullframe", B, ph, pw) / mean / 1e9))
        del g, keep
    return out


def k1_sweep(batches=(8, 16, 64, 256), ph=50, pw=50, iters=20, device="cuda:0"):
    """K1 at the per-rank batch sizes of the BASELINE configs (strong scaling: 64 / 8 ranks = 8; config 2: 16) and beyond."""
    dev = torch.device(device)
    out = []
    patch = torch.rand(3, ph, pw, device=dev)
    for B in batches:
        img = torch.from_numpy(synthetic.synth_images(1234, min(B, 64), "noise")).to(dev)
        if B > img.shape[0]:
            img = img.repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
        xy_n, th_n = random_params(B, ph, pw, 42)
        xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
        mean, med, mn = _time(lambda: ops.patch_apply_fwd(img, patch, xy, th, True), iters)
        nb = algo_bytes("K1", B, ph, pw)
        out.append(dict(B=B, mean_us=mean * 1e6, min_us=mn * 1e6, algo_bytes=nb, achieved_GBs=nb / mean / 1e9, frac_of_8TBs=nb / mean / 1e9 / HBM_PEAK_GBS))
    return out


def rank_shapes(iters=20, device="cuda:0"):
    """Kernel numbers at the per-rank shapes of the multi-GPU BASELINE configs: config 3 strong scaling (B=8, 50x50, UADA_DDP), config 4
    (TMA, B=8, all 7 DoF) and config 5 (UPA, resize_patch=True, 3x100x100 base, B=4: resize + K1/K2 with per-image patches)."""
    import random

    from .transform import RandomPatchTransform

    dev = torch.device(device)
    res = {}
    s8 = kernel_suite(8, 50, 50, iters=iters, device=device)
    res["cfg3_B8_50x50"] = {k: v for k, v in s8.items() if not k.startswith("K2e")}
    # config 4 (TMA, 32 over 4 ranks -> B=8 per rank, all 7 DoF + EOS = 64 labelled rows, CE to the target tokens: full-row gradient)
    from .labels import tma_target_labels, tma_target_tokens

    _, lab4, _ = synthetic.synth_text_batch(4242, 8)
    lab4 = tma_target_labels(lab4, tma_target_tokens(np.zeros(7), list(range(7)))).to(dev)
    R4 = int((lab4[:, 1:] != -100).sum())
    lg4 = (torch.randn(R4, 32064, device=dev) * 2).to(torch.bfloat16)
    g4 = torch.empty_like(lg4)
    rm4 = ops.LossRowMap(lab4)
    mean, med, mn = _time(lambda: ops.loss_rows_fwd_bwd(lg4, rm4, ops.LOSS_CE, grad_kind=ops.GRAD_FULL, grad=g4), iters)
    nb = algo_bytes("K3", 8, rows=R4, esize=2)
    res["cfg4_tma_B8"] = {"K3_full_rows_fwd_bwd": dict(mean_us=mean * 1e6, min_us=mn * 1e6, algo_bytes=nb, achieved_GBs=nb / mean / 1e9,
                                                       frac_of_8TBs=nb / mean / 1e9 / HBM_PEAK_GBS, rows=R4),
                         "note": "K1/K2/K4 as in cfg3_B8_50x50 (same per-rank batch and patch)"}
    # config 5: the whole resized forward/backward of the patch operator (4 launches + fixed-order reduce), B = 4
    B = 4
    img = torch.from_numpy(synthetic.synth_images(5, B, "noise")).to(dev)
    patch = torch.rand(3, 100, 100, device=dev)
    st_r, st_n = random.getstate(), np.random.get_state()
    random.seed(42)
    np.random.seed(42)
    sizes, xy_n, th_n = RandomPatchTransform("cpu", True)._draw_resized(B, 100, 100, True)
    random.setstate(st_r)
    np.random.set_state(st_n)
    xy, th = torch.from_numpy(xy_n).to(dev), torch.from_numpy(th_n).to(dev)
    pdesc_n, total = ops.make_pdesc(sizes)
    pdesc = torch.from_numpy(pdesc_n).to(dev)
    max_hw = (int(pdesc_n[:, 0].max()), int(pdesc_n[:, 1].max()))
    packed = ops.patch_resize_fwd(patch, pdesc, total)
    _, keep = ops.patch_apply_fwd_multi(img, packed, pdesc, max_hw, xy, th, True)
    g = synthetic.synth_upstream_grad(7, B).to(dev)
    gp = ops.patch_grad_gather_multi(g, packed, pdesc, max_hw, xy, th, keep, True)
    kept_px = int(sum(int(h) * int(w) for h, w in sizes))
    dy0 = (torch.randn(B, 256, 1024, device=dev) * 0.1).to(torch.bfloat16)
    dy1 = (torch.randn(B, 256, 1152, device=dev) * 0.1).to(torch.bfloat16)
    wp0 = ops.pack_embed_weights((torch.randn(588, 1024, device=dev) * 0.05).to(torch.bfloat16))
    wp1 = ops.pack_embed_weights((torch.randn(588, 1152, device=dev) * 0.05).to(torch.bfloat16))
    ntile = int(sum(((int(h) + 13) // 14 + 1) * ((int(w) + 13) // 14 + 1) for h, w in sizes))  # ~tiles under the (warped) patches
    _, _, keep_t, tflags = ops.patch_apply_fwd_tiles(img, packed, xy, th, True, pdesc=pdesc, max_hw=max_hw)
    c5 = {}
    for name, fn, nb in (
        ("K1t_patch_apply_fwd_tiles_multi", lambda: ops.patch_apply_fwd_tiles(img, packed, xy, th, True, pdesc=pdesc, max_hw=max_hw), B * (150528 + 602112) + 4 * total),
        ("K2et_patch_embed_grad_gather_multi_tiles",
         lambda: ops.patch_embed_grad_gather_multi_tiles(dy0, dy1, wp0, wp1, packed, pdesc, max_hw, xy, th, keep_t, tflags, True),
         ntile * (1024 + 1152) * 2 + 588 * (1024 + 1152) * 2 + 4 * total),
        ("K2e_patch_embed_grad_gather_multi",
         lambda: ops.patch_embed_grad_gather_multi(dy0, dy1, wp0, wp1, packed, pdesc, max_hw, xy, th, keep, True),
         ntile * (1024 + 1152) * 2 + 588 * (1024 + 1152) * 2 + 4 * total),
        ("K0_patch_resize_fwd", lambda: ops.patch_resize_fwd(patch, pdesc, total), 4 * 3 * 100 * 100 + 4 * total),
        ("K1_patch_apply_fwd_multi", lambda: ops.patch_apply_fwd_multi(img, packed, pdesc, max_hw, xy, th, True), B * (150528 + 602112) + 4 * total),
        ("K2_patch_grad_gather_multi", lambda: ops.patch_grad_gather_multi(g, packed, pdesc, max_hw, xy, th, keep, True), 12 * kept_px + 4 * total),
        ("K0_patch_resize_bwd", lambda: ops.patch_resize_bwd(gp, pdesc, 100, 100), 4 * total + 4 * 3 * 100 * 100),
    ):
        mean, med, mn = _time(fn, iters)
        c5[name] = dict(mean_us=mean * 1e6, min_us=mn * 1e6, algo_bytes=nb, achieved_GBs=nb / mean / 1e9, frac_of_8TBs=nb / mean / 1e9 / HBM_PEAK_GBS)
    c5["sizes"] = [[int(h), int(w)] for h, w in sizes]
    res["cfg5_B4_resize100"] = c5
    return res


def device_copy_bandwidth(nbytes=512 * 1024 * 1024, iters=5, device="cuda:0"):
    """Measured device-to-device copy bandwidth (read + write bytes per second) of this box: the practical HBM ceiling the
    roofline fractions can also be read against. 512 MiB > the 256 MiB Infinity Cache."""
    dev = torch.device(device)
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        dst.copy_(src)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e-3)
    del src, dst
    return 2.0 * nbytes / float(np.median(ts)) / 1e9
