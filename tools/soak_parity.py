#!/usr/bin/env python3
"""Randomised parity soak of the HIP patch operators against the plain-C oracle (GPU box; not part of the test suite).

    python tools/soak_parity.py --seconds 240 --seed 1

Every case draws a patch size (1..224 per side), a batch, placements biased towards the frame edges, and either the reference's
rotation+shear transforms or a general affine (scale 0.3..3, any rotation, translation), then checks
  K1  : whole bf16 tensor and keep bits BIT-EXACT (uniform patch and per-image patches), in the planar AND the tile-major form (both patch-embed
        GEMM operands, keep words, tile flags),
  K2  : <= 3e-6 of the gradient scale with the stored mask and with the recomputed mask, bitwise repeatable,
  K2' : <= 5e-3 (one bf16 ulp: rounding flips of the tile gradients) against the oracle fed by an fp32 host matmul, random tower widths; the
        tile-major-mask form bitwise the planar-mask form; the step epilogue (final sum + AdamW inside it) bitwise the separate launches,
  K0  : resize forward bit-exact, adjoint <= 2e-6,
  K5  : eval-time paste byte-exact.
Prints one line per failure (with the seed that reproduces it) and a summary; exit code 1 on any failure.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import c_oracle, ref_port  # noqa: E402
from roboticattack_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return (x.to(dtype) if dtype is not None else x).to(DEV)


def bits(x):
    return x.view(torch.int16).cpu().numpy().view(np.uint16)


def draw_theta(rs, B, general):
    th = np.zeros((B, 2, 3), np.float32)
    for b in range(B):
        if rs.rand() < 0.15:
            m = np.eye(3, dtype=np.float32)
        elif general:
            a = rs.uniform(-np.pi, np.pi)
            sx, sy = rs.uniform(0.3, 3.0), rs.uniform(0.3, 3.0)
            sh = rs.uniform(-0.5, 0.5)
            lin = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]) @ np.array([[sx, sh], [0, sy]])
            m = np.eye(3, dtype=np.float32)
            m[:2, :2] = lin
            m[:2, 2] = rs.uniform(-0.8, 0.8, 2)
        else:
            m = np.dot(ref_port.shear_matrix(rs.uniform(-0.2, 0.2), rs.uniform(-0.2, 0.2)), ref_port.rotation_matrix(rs.uniform(-30, 30)))
        th[b] = m[:2]
    return th


def draw_xy(rs, sizes):
    xy = np.zeros((len(sizes), 2), np.int32)
    for b, (h, w) in enumerate(sizes):
        xy[b] = (rs.randint(0, 224 - w + 1), rs.randint(0, 224 - h + 1))
        if rs.rand() < 0.3:
            xy[b, 0] = rs.choice([0, 224 - w])
        if rs.rand() < 0.3:
            xy[b, 1] = rs.choice([0, 224 - h])
    return xy


def one_case(seed):
    rs = np.random.RandomState(seed)
    fails = []
    side = lambda: int(rs.choice([rs.randint(1, 225), rs.randint(20, 140), 50]))
    ph, pw = side(), side()
    B = int(rs.choice([1, 2, 3, 5, 9, 17, 40])) if ph * pw < 150 * 150 else int(rs.choice([1, 2, 3]))
    geo = int(rs.rand() < 0.85)
    mm = int(rs.rand() < 0.5) if not geo else 0  # `canvas != -100` is defined for the un-warped paste only
    general = rs.rand() < 0.3
    imgs = synthetic.synth_images(seed % 1000, B, "noise" if rs.rand() < 0.5 else "smooth")
    patch = rs.rand(3, ph, pw).astype(np.float32)
    if rs.rand() < 0.2:
        patch = (patch * 60 - 40).astype(np.float32)  # values around the -20 threshold: the mask depends on the patch itself
    theta = draw_theta(rs, B, general)
    xy = draw_xy(rs, [(ph, pw)] * B)
    g = (synthetic.synth_upstream_grad(seed % 977, B).float() * float(10 ** rs.uniform(-6, 3))).to(torch.bfloat16)
    tag = f"seed={seed} B={B} patch={ph}x{pw} geo={geo} mask_mode={mm} general={general}"
    # ---- uniform patch ----
    out, keep = ops.patch_apply_fwd(t(imgs), t(patch), t(xy, torch.int32), t(theta.reshape(-1, 6)), bool(geo), mm)
    _, ob, ok = c_oracle.patch_apply_fwd(imgs, patch, xy, theta, geo, mm)
    if not np.array_equal(bits(out), ob):
        fails.append(f"K1 out   {tag}: {int((bits(out) != ob).sum())} values differ")
    if not np.array_equal(np.unpackbits(keep.cpu().numpy(), axis=-1, bitorder="little"), ok):
        fails.append(f"K1 keep  {tag}")
    # ---- K1 in tile-major form (what the attack step runs): both GEMM operands == im2col of the oracle's bf16 tensor, keep words == the
    #      oracle's mask bits, tile flags == "any kept pixel in the tile" ----
    t0, t1, keep_t, tflags = ops.patch_apply_fwd_tiles(t(imgs), t(patch), t(xy, torch.int32), t(theta.reshape(-1, 6)), bool(geo), mm)
    ob_t = torch.from_numpy(ob.view(np.int16)).view(B, 6, 224, 224)
    im2col = lambda x3: x3.reshape(B, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(B, 256, 588)
    if not (torch.equal(t0.cpu().view(torch.int16), im2col(ob_t[:, :3])) and torch.equal(t1.cpu().view(torch.int16), im2col(ob_t[:, 3:]))):
        fails.append(f"K1 tiles out  {tag}")
    okb = ok.reshape(B, 3, 16, 14, 16, 14)                                                   # [b,c,ty,y,tx,x] mask bits of the oracle
    kw = (okb.astype(np.uint32) << np.arange(14, dtype=np.uint32)).sum(axis=5).transpose(0, 1, 2, 4, 3).reshape(B, 3, 256, 14).astype(np.uint16)
    if not np.array_equal(keep_t.cpu().numpy().view(np.uint16), kw) or not np.array_equal(tflags.cpu().numpy() != 0, (kw != 0).any(axis=(1, 3))):
        fails.append(f"K1 tiles keep {tag}")
    # general affines can map many output pixels onto one texel: the reference's fp32 scan-order accumulation is then itself the
    # dominant error, so those cases are checked against the same fp32 products accumulated in fp64
    og = c_oracle.patch_grad(bits(g), patch, xy, theta, geo, mm, f64=general)
    sc = max(np.abs(og).max(), 1e-30)
    args = (g.to(DEV), t(patch), t(xy, torch.int32), t(theta.reshape(-1, 6)))
    g1 = ops.patch_grad_gather(*args, keep, bool(geo), mm)
    g2 = ops.patch_grad_gather(*args, None, bool(geo), mm)
    if not (np.abs(g1.cpu().numpy() - og).max() <= 3e-6 * sc):
        fails.append(f"K2       {tag}: rel err {np.abs(g1.cpu().numpy() - og).max() / sc:.3e}")
    if not torch.equal(g1, g2) or not torch.equal(g1, ops.patch_grad_gather(*args, keep, bool(geo), mm)):
        fails.append(f"K2 repeat/mask {tag}")
    # ---- K2': patch-embed backward on the tiles under the patch + gather, against the oracle fed by an fp32 host matmul ----
    if mm == 0 and rs.rand() < 0.3:
        D0, D1 = int(rs.choice([64, 128, 192, 320])), int(rs.choice([64, 128, 256]))
        if rs.rand() < 0.1:
            D0 = 1216  # too wide for the LDS-resident variant
        gen = torch.Generator(device=DEV).manual_seed(seed % 100003)
        dy = [(torch.randn(B, 256, D, device=DEV, generator=gen) * float(10 ** rs.uniform(-4, 1))).to(torch.bfloat16) for D in (D0, D1)]
        w = [(torch.randn(D, 588, device=DEV, generator=gen) * 0.05).to(torch.bfloat16) for D in (D0, D1)]
        fused = ops.patch_embed_grad_gather(dy[0], dy[1], ops.pack_embed_weights(w[0].t().contiguous()), ops.pack_embed_weights(w[1].t().contiguous()),
                                            t(patch), t(xy, torch.int32), t(theta.reshape(-1, 6)), keep, bool(geo)).cpu().numpy()
        # the production forms: K2' fed by the tile-major mask (bitwise the planar-mask form) and its final sum left to the step epilogue,
        # with the optimiser applied inside it (bitwise vaa_patch_update on the same gradient)
        wps = (ops.pack_embed_weights(w[0].t().contiguous()), ops.pack_embed_weights(w[1].t().contiguous()))
        pa = (t(patch), t(xy, torch.int32), t(theta.reshape(-1, 6)))
        ft = ops.patch_embed_grad_gather_tiles(dy[0], dy[1], *wps, *pa, keep_t, tflags, bool(geo))
        if not np.array_equal(ft.cpu().numpy(), fused):
            fails.append(f"K2' tiles {tag} D={D0}+{D1}: differs from the planar-mask form")
        parts = ops.patch_embed_grad_gather_tiles(dy[0], dy[1], *wps, *pa, keep_t, tflags, bool(geo), defer_reduce=True)
        n_el = 3 * ph * pw
        msg, sc8 = torch.zeros(n_el + 4, device=DEV), torch.arange(8, dtype=torch.float32, device=DEV)
        p_a, m_a, v_a = (torch.from_numpy(x).to(DEV) for x in (rs.rand(3, ph, pw).astype(np.float32), (rs.rand(3, ph, pw) * 1e-3).astype(np.float32),
                                                                (rs.rand(3, ph, pw) * 1e-6).astype(np.float32)))
        p_b, m_b, v_b = p_a.clone(), m_a.clone(), v_a.clone()
        ops.step_epilogue(parts, msg, sc8, update=dict(patch=p_b, m=m_b, v=v_b, mode=ops.OPT_ADAMW_HF, lr=2e-3, step=3))
        ops.patch_update(p_a, ft, m_a, v_a, ops.OPT_ADAMW_HF, 2e-3, 3)
        if not (torch.equal(msg[:n_el].view_as(ft), ft) and torch.equal(p_a, p_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b)):
            fails.append(f"epilogue {tag} D={D0}+{D1}: sum / fused update differ from the separate launches")
        fold = lambda d, ww: (d.float().cpu() @ ww.float().cpu()).to(torch.bfloat16).view(B, 16, 16, 3, 14, 14).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 224, 224)
        gcat = torch.cat([fold(dy[0], w[0]), fold(dy[1], w[1])], dim=1).contiguous()
        oge = c_oracle.patch_grad(bits(gcat), patch, xy, theta, geo, 0, f64=general)
        # one bf16 ulp (2^-8) of a tile-gradient element flips where the two GEMMs' fp32 sums straddle a rounding boundary; with a single
        # image a texel can be one pixel's contribution, so the bound is an ulp of the largest gradient, not the 2e-3 of the batched tests
        if not (np.abs(fused - oge).max() <= 5e-3 * max(np.abs(oge).max(), 1e-30) + 1e-30):
            fails.append(f"K2'      {tag} D={D0}+{D1}: rel err {np.abs(fused - oge).max() / max(np.abs(oge).max(), 1e-30):.3e}")
    # ---- per-image patches + resize (config 5 path), mask rule lt-20 ----
    if rs.rand() < 0.5:
        bh, bw = int(rs.randint(2, 120)), int(rs.randint(2, 120))
        base = rs.rand(3, bh, bw).astype(np.float32)
        sizes = np.array([[max(1, int(bh * s)), max(1, int(bw * s))] for s in rs.uniform(0.5, 1.8, B)], np.int32).clip(1, 224)
        pdesc_n, total = ops.make_pdesc(sizes)
        packed = ops.patch_resize_fwd(t(base), t(pdesc_n), total)
        o_packed = c_oracle.patch_resize_fwd(base, pdesc_n, total)
        if not np.array_equal(packed.cpu().numpy(), o_packed):
            fails.append(f"K0 fwd   {tag} base={bh}x{bw} sizes={sizes.tolist()}")
        xy2 = draw_xy(rs, sizes)
        mh = (int(sizes[:, 0].max()), int(sizes[:, 1].max()))
        o2, k2 = ops.patch_apply_fwd_multi(t(imgs), t(o_packed), t(pdesc_n), mh, t(xy2, torch.int32), t(theta.reshape(-1, 6)), bool(geo), 0)
        _, ob2, ok2 = c_oracle.patch_apply_fwd_multi(imgs, o_packed, pdesc_n, xy2, theta, geo, 0)
        if not np.array_equal(bits(o2), ob2) or not np.array_equal(np.unpackbits(k2.cpu().numpy(), axis=-1, bitorder="little"), ok2):
            fails.append(f"K1 multi {tag} sizes={sizes.tolist()}")
        gp = ops.patch_grad_gather_multi(g.to(DEV), t(o_packed), t(pdesc_n), mh, t(xy2, torch.int32), t(theta.reshape(-1, 6)), k2, bool(geo), 0).cpu().numpy()
        if general:
            o_gp = np.zeros(total, np.float32)
            for b, (h, w, off, _z) in enumerate(pdesc_n):
                o_gp[off : off + 3 * h * w] = c_oracle.patch_grad(bits(g)[b : b + 1], o_packed[off : off + 3 * h * w].reshape(3, h, w), xy2[b : b + 1],
                                                                  theta[b : b + 1], geo, 0, f64=True).ravel()
        else:
            o_gp = c_oracle.patch_grad_multi(bits(g), o_packed, pdesc_n, xy2, theta, geo, 0)
        for (h, w, off, _z) in pdesc_n:
            a, b = gp[off : off + 3 * h * w], o_gp[off : off + 3 * h * w]
            if not (np.abs(a - b).max() <= 3e-6 * max(np.abs(b).max(), 1e-30)):
                fails.append(f"K2 multi {tag} image {h}x{w}: rel err {np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.3e}")
        if not general and rs.rand() < 0.4:  # K2' in per-image mode against the oracle fed by an fp32 host matmul
            D0, D1 = int(rs.choice([64, 128, 192])), int(rs.choice([64, 128]))
            gen = torch.Generator(device=DEV).manual_seed(seed % 100019)
            dy = [(torch.randn(B, 256, D, device=DEV, generator=gen) * 0.1).to(torch.bfloat16) for D in (D0, D1)]
            w = [(torch.randn(D, 588, device=DEV, generator=gen) * 0.05).to(torch.bfloat16) for D in (D0, D1)]
            fm = ops.patch_embed_grad_gather_multi(dy[0], dy[1], ops.pack_embed_weights(w[0].t().contiguous()), ops.pack_embed_weights(w[1].t().contiguous()),
                                                   t(o_packed), t(pdesc_n), mh, t(xy2, torch.int32), t(theta.reshape(-1, 6)), k2, bool(geo)).cpu().numpy()
            fold = lambda d, ww: (d.float().cpu() @ ww.float().cpu()).to(torch.bfloat16).view(B, 16, 16, 3, 14, 14).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 224, 224)
            gcat = torch.cat([fold(dy[0], w[0]), fold(dy[1], w[1])], dim=1).contiguous()
            om = c_oracle.patch_grad_multi(bits(gcat), o_packed, pdesc_n, xy2, theta, geo, 0)
            # the same gather of the MAGNITUDES of the pixel gradients: what one bf16 rounding step (2^-8) of every contribution can add up to. A per-image
            # gradient is the sum of two towers' contributions that may cancel; 5e-3 of the image's largest element covers one step of that element only
            # (seed 71004666: 5.6e-3 on one 98x181 image)
            omag = c_oracle.patch_grad_multi(bits(gcat.abs()), o_packed, pdesc_n, xy2, theta, geo, 0)
            for (h, w_, off, _z) in pdesc_n:
                a, b = fm[off : off + 3 * h * w_], om[off : off + 3 * h * w_]
                if not (np.abs(a - b) <= 5e-3 * max(np.abs(b).max(), 1e-30) + 2.0 ** -8 * np.abs(omag[off : off + 3 * h * w_]) + 1e-30).all():
                    fails.append(f"K2' multi {tag} image {h}x{w_} D={D0}+{D1}: rel err {np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.3e}")
        gb = ops.patch_resize_bwd(t(o_gp), t(pdesc_n), bh, bw).cpu().numpy()
        o_gb = c_oracle.patch_resize_bwd(o_gp, pdesc_n, bh, bw)
        if not (np.abs(gb - o_gb).max() <= 2e-6 * max(np.abs(o_gb).max(), 1e-30)):
            fails.append(f"K0 bwd   {tag} base={bh}x{bw}: rel err {np.abs(gb - o_gb).max() / max(np.abs(o_gb).max(), 1e-30):.3e}")
    # ---- eval-time paste ----
    if rs.rand() < 0.3 and patch.min() >= 0:
        geo_b = rs.rand(B) < 0.7
        got = ops.patch_apply_eval(t(imgs), t(np.clip(patch, 0, 1)), t(xy, torch.int32), t(theta.reshape(-1, 6)), t(geo_b.astype(np.int32))).cpu().numpy()
        if not np.array_equal(got, c_oracle.patch_apply_eval(imgs, np.clip(patch, 0, 1), xy, theta, geo_b)):
            fails.append(f"K5       {tag}")
    return fails


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--case", type=int, default=None, help="run ONE case with this full case seed (as printed in a FAIL line) and exit")
    a = ap.parse_args()
    ops.device_check()
    if a.case is not None:
        bad = one_case(a.case)
        print("\n".join("FAIL " + b for b in bad) or f"case {a.case}: ok")
        sys.exit(1 if bad else 0)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < a.seconds:
        bad += one_case(a.seed * 1000003 + n)
        n += 1
    for b in bad:
        print("FAIL", b)
    print(f"soak: {n} cases in {time.time() - t0:.0f} s, {len(bad)} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
