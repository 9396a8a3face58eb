/*
 * ORACLE (test infrastructure, NOT product code): plain-C restatement of the UADA/UPA/TMA hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. It is
 * the deterministic scalar checker the HIP kernels are compared with (mask indices bit-exact,
 * values to tolerance). Build: `make -C oracle` (gcc -O2 -ffp-contract=off; FMAs are explicit).
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against vectors recorded by
 * tools/gen_golden.py from the reference's own Python functions (tests/golden/ *.npz).
 *
 * Reference citations are relative to /root/reference (read-only, absent on the GPU box).
 * Numerics of the warp follow SURVEY.md Appendix B (torch CPU affine_grid/grid_sample, verified
 * against the fixtures):
 *   base grid   : torch.linspace(-1,1,224)*(223/224)                       [3p torch affine_grid]
 *   grid        : gx = fma(by[i],t01, bx[j]*t00) + t02                     [3p torch bmm 3-term product]
 *   unnormalise : ix = fma(gx+1, 112, -0.5), clamp to [0,223] (padding 'border') [3p torch grid_sample;
 *                 the two-rounding form ((gx+1)*224-1)/2 differs from torch CPU on ~1% of pixels by 1 ulp]
 *   bilinear    : out = fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw)))
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IMG 224
#define NPIX (IMG * IMG)

/* mask_mode: 0 = `torch.where(canvas < -20, im, canvas)`  (appply_random_transform.py:131)
 *            1 = `torch.where(canvas != -100, canvas, im)` (appply_random_transform.py:153,179) */

static float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static uint16_t f32_to_bf16_rne(float f) { /* torch .to(torch.bfloat16): round-to-nearest-even, NaN quieted */
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}

/* torch.linspace(-1, 1, 224) (float) then * (223/224): F.affine_grid(align_corners=False) base grid
 * (appply_random_transform.py:98). linspace: i < 112 -> start + step*i ; else end - step*(223-i), one rounding each. */
static void base_grid(float* b) {
    const float step = (1.0f - (-1.0f)) / 223.0f;
    for (int i = 0; i < IMG; ++i) {
        float lin = (i < IMG / 2) ? fmaf((float)i, step, -1.0f) : fmaf(-(float)(IMG - 1 - i), step, 1.0f);
        b[i] = (lin * 223.0f) / 224.0f;
    }
}

typedef struct {
    int x0, y0;          /* floor of the clamped sample position */
    float nw, ne, sw, se; /* bilinear weights (appply_random_transform.py:100 -> torch grid_sample) */
} samp_t;

static samp_t sample_pos(const float* bgrid, const float* th /*2x3 row-major*/, int i, int j) {
    samp_t s;
    float bx = bgrid[j], by = bgrid[i];
    float gx = fmaf(by, th[1], bx * th[0]) + th[2];
    float gy = fmaf(by, th[4], bx * th[3]) + th[5];
    float ix = fmaf(gx + 1.0f, 112.0f, -0.5f); /* torch CPU kernel contracts (g+1)*(size/2)-0.5 into one FMA */
    float iy = fmaf(gy + 1.0f, 112.0f, -0.5f);
    ix = fminf(223.0f, fmaxf(ix, 0.0f));
    iy = fminf(223.0f, fmaxf(iy, 0.0f));
    float xw = floorf(ix), yn = floorf(iy);
    float w = ix - xw, e = 1.0f - w, n = iy - yn, so = 1.0f - n;
    s.x0 = (int)xw;
    s.y0 = (int)yn;
    s.nw = so * e;
    s.ne = so * w;
    s.sw = n * e;
    s.se = n * w;
    return s;
}

static inline float canvas_at(const float* patch, int c, int ph, int pw, int px, int py, int xx, int yy) {
    /* canvas = -100 everywhere, patch pasted at (px,py) (appply_random_transform.py:111,125); OOB corner reads 0 */
    if (xx < 0 || xx >= IMG || yy < 0 || yy >= IMG) return 0.0f;
    int u = xx - px, v = yy - py;
    if (u >= 0 && u < pw && v >= 0 && v < ph) return patch[(c * ph + v) * pw + u];
    return -100.0f;
}

static inline float sample_canvas(const float* patch, int c, int ph, int pw, int px, int py, const samp_t* s) {
    float vnw = canvas_at(patch, c, ph, pw, px, py, s->x0, s->y0);
    float vne = canvas_at(patch, c, ph, pw, px, py, s->x0 + 1, s->y0);
    float vsw = canvas_at(patch, c, ph, pw, px, py, s->x0, s->y0 + 1);
    float vse = canvas_at(patch, c, ph, pw, px, py, s->x0 + 1, s->y0 + 1);
    return fmaf(vse, s->se, fmaf(vsw, s->sw, fmaf(vne, s->ne, vnw * s->nw)));
}

static inline int keep_rule(float canvas_v, int mask_mode) {
    return mask_mode == 0 ? !(canvas_v < -20.0f) : (canvas_v != -100.0f);
}

/* K1: appply_random_transform.py:104-136 (+ `.to(torch.bfloat16)` of UADA.py:142 when out_bf16 != NULL).
 * img_u8 [B,224,224,3] HWC; patch [3,ph,pw]; xy [B,2] (x,y); theta [B,6]; outputs may be NULL. */
void vaa_oracle_patch_apply_fwd(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, int B,
                                int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6,
                                float* out_f32, uint16_t* out_bf16, uint8_t* keep /*[B,3,224*224] 0/1*/) {
    float bgrid[IMG];
    base_grid(bgrid);
    for (int b = 0; b < B; ++b) {
        const int px = xy[2 * b], py = xy[2 * b + 1];
        const float* th = theta + 6 * b;
        for (int i = 0; i < IMG; ++i)
            for (int j = 0; j < IMG; ++j) {
                samp_t s = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
                if (geometry) s = sample_pos(bgrid, th, i, j);
                for (int c = 0; c < 3; ++c) {
                    float cv = geometry ? sample_canvas(patch, c, ph, pw, px, py, &s)
                                        : canvas_at(patch, c, ph, pw, px, py, j, i);
                    int k = keep_rule(cv, mask_mode);
                    float im = (float)img_u8[((size_t)(b * IMG + i) * IMG + j) * 3 + c] / 255.0f; /* ToTensor :108 */
                    float v = k ? cv : im;
                    float o0 = (v - mean6[c]) / std6[c];         /* :132 */
                    float o1 = (v - mean6[c + 3]) / std6[c + 3]; /* :133 */
                    size_t o = ((size_t)(b * 6 + c) * IMG + i) * IMG + j;
                    if (out_f32) { out_f32[o] = o0; out_f32[o + (size_t)3 * NPIX] = o1; }
                    if (out_bf16) { out_bf16[o] = f32_to_bf16_rne(o0); out_bf16[o + (size_t)3 * NPIX] = f32_to_bf16_rne(o1); }
                    if (keep) keep[((size_t)(b * 3 + c)) * NPIX + i * IMG + j] = (uint8_t)k;
                }
            }
    }
}

/* K2: autograd backward of K1 w.r.t. the patch (implicit at UADA.py:148), closed form:
 *   G_c = (g_c/std0_c + g_{c+3}/std1_c) * [kept];  d canvas += bilinear-scatter(G);  d patch = sum_b d canvas[rect_b].
 * gout_bf16 [B,6,224,224]; gpatch [3,ph,pw] overwritten. fp32 accumulation in (i,j) scan order like torch CPU. */
void vaa_oracle_patch_grad(const uint16_t* gout_bf16, const float* patch, const int32_t* xy, const float* theta, int B,
                           int ph, int pw, int geometry, int mask_mode, const float* std6, float* gpatch) {
    float bgrid[IMG];
    base_grid(bgrid);
    const int n = 3 * ph * pw;
    float* acc = (float*)malloc(sizeof(float) * n);
    memset(gpatch, 0, sizeof(float) * n);
    for (int b = B - 1; b >= 0; --b) { /* autograd visits the last-created branch first */
        memset(acc, 0, sizeof(float) * n);
        const int px = xy[2 * b], py = xy[2 * b + 1];
        const float* th = theta + 6 * b;
        for (int i = 0; i < IMG; ++i)
            for (int j = 0; j < IMG; ++j) {
                samp_t s = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
                if (geometry) s = sample_pos(bgrid, th, i, j);
                for (int c = 0; c < 3; ++c) {
                    float cv = geometry ? sample_canvas(patch, c, ph, pw, px, py, &s)
                                        : canvas_at(patch, c, ph, pw, px, py, j, i);
                    if (!keep_rule(cv, mask_mode)) continue;
                    size_t o = ((size_t)(b * 6 + c) * IMG + i) * IMG + j;
                    float G = bf16_to_f32(gout_bf16[o]) / std6[c] + bf16_to_f32(gout_bf16[o + (size_t)3 * NPIX]) / std6[c + 3];
                    if (!geometry) {
                        int u = j - px, v = i - py;
                        if (u >= 0 && u < pw && v >= 0 && v < ph) acc[(c * ph + v) * pw + u] += G;
                        continue;
                    }
                    const int cx[4] = {s.x0, s.x0 + 1, s.x0, s.x0 + 1};
                    const int cy[4] = {s.y0, s.y0, s.y0 + 1, s.y0 + 1};
                    const float wt[4] = {s.nw, s.ne, s.sw, s.se};
                    for (int q = 0; q < 4; ++q) {
                        int u = cx[q] - px, v = cy[q] - py;
                        if (cx[q] >= IMG || cy[q] >= IMG) continue;
                        if (u >= 0 && u < pw && v >= 0 && v < ph) acc[(c * ph + v) * pw + u] += G * wt[q];
                    }
                }
            }
        for (int k = 0; k < n; ++k) gpatch[k] += acc[k];
    }
    free(acc);
}

/* The same gradient with the fp32 PRODUCTS of the reference (G, G*w) but fp64 ACCUMULATION: the exact sum up to 1e-16, used where the
 * fp32 scan-order accumulation above is itself the dominant error (transforms that map many output pixels onto one texel). */
void vaa_oracle_patch_grad_f64(const uint16_t* gout_bf16, const float* patch, const int32_t* xy, const float* theta, int B,
                           int ph, int pw, int geometry, int mask_mode, const float* std6, float* gpatch) {
    float bgrid[IMG];
    base_grid(bgrid);
    const int n = 3 * ph * pw;
    double* acc = (double*)malloc(sizeof(double) * n);
    double* tot = (double*)calloc(n, sizeof(double));
    memset(gpatch, 0, sizeof(float) * n);
    for (int b = B - 1; b >= 0; --b) { /* autograd visits the last-created branch first */
        memset(acc, 0, sizeof(double) * n);
        const int px = xy[2 * b], py = xy[2 * b + 1];
        const float* th = theta + 6 * b;
        for (int i = 0; i < IMG; ++i)
            for (int j = 0; j < IMG; ++j) {
                samp_t s = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
                if (geometry) s = sample_pos(bgrid, th, i, j);
                for (int c = 0; c < 3; ++c) {
                    float cv = geometry ? sample_canvas(patch, c, ph, pw, px, py, &s)
                                        : canvas_at(patch, c, ph, pw, px, py, j, i);
                    if (!keep_rule(cv, mask_mode)) continue;
                    size_t o = ((size_t)(b * 6 + c) * IMG + i) * IMG + j;
                    float G = bf16_to_f32(gout_bf16[o]) / std6[c] + bf16_to_f32(gout_bf16[o + (size_t)3 * NPIX]) / std6[c + 3];
                    if (!geometry) {
                        int u = j - px, v = i - py;
                        if (u >= 0 && u < pw && v >= 0 && v < ph) acc[(c * ph + v) * pw + u] += (double)G;
                        continue;
                    }
                    const int cx[4] = {s.x0, s.x0 + 1, s.x0, s.x0 + 1};
                    const int cy[4] = {s.y0, s.y0, s.y0 + 1, s.y0 + 1};
                    const float wt[4] = {s.nw, s.ne, s.sw, s.se};
                    for (int q = 0; q < 4; ++q) {
                        int u = cx[q] - px, v = cy[q] - py;
                        if (cx[q] >= IMG || cy[q] >= IMG) continue;
                        if (u >= 0 && u < pw && v >= 0 && v < ph) acc[(c * ph + v) * pw + u] += (double)(G * wt[q]);
                    }
                }
            }
        for (int k = 0; k < n; ++k) tot[k] += acc[k];
    }
    for (int k = 0; k < n; ++k) gpatch[k] = (float)tot[k];
    free(acc);
    free(tot);
}

/* ---------------------------------------------------------------------------------------------
 * K3 losses on the labelled rows of logits [B,S,V] f32 with labels [B,L] (S = 256 + L).
 * Row (b,k), k in [0,L-1): model position p = S-L+k predicts labels[b,k+1] (UADA.py:382-386; HF shift).
 *   mode 0 UADA      : w^2*mean_R((r-t)^2) + 1/CE        UADA.py:145-148, 381-406
 *   mode 1 UADA_DDP  : w^2*mean_R((r-t)^2)               UADA_ddp.py:99-124, 203-206
 *   mode 2 UPA       : alpha*mean_b(cos+1) + beta/(mean_b||e'-l'|| + 1e-3)   UPA.py:367-387
 *   mode 3 TMA / CE  : scale*CE                          TMA.py:148 (scale = 1/accumulate_steps)
 *   r = sum_k softmax(z[31744:32000])_k*(k+1)/256, t = (label > 31872) ? 0 : 1   (A-D10 reproduced)
 *   CE = mean over rows with label != -100 of (logsumexp(z) - z[label])           [3p HF Llama loss]
 * params: [w, alpha, beta, scale].  scalars out: [total, ce, mse, aux0(angle), aux1(dist), n_ce_rows, n_act_rows, 0].
 * glog (may be NULL): [B,S,V] f32, written ONLY on labelled rows (caller zero-fills).
 * --------------------------------------------------------------------------------------------- */
void vaa_oracle_loss(const float* logits, const int64_t* labels, int B, int S, int L, int V, int mode, const float* params,
                     float* scalars, float* glog) {
    const int A0 = 31744, NA = 256;
    const double w = params[0], alpha = params[1], beta = params[2], scale = params[3];
    int nrows = 0, nact = 0;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k + 1 < L; ++k) {
            int64_t lab = labels[(size_t)b * L + k + 1];
            if (lab != -100) ++nrows;
            if (lab > 2) ++nact;
        }
    double ce_sum = 0.0, mse_sum = 0.0;
    double* lse = (double*)malloc(sizeof(double) * (size_t)B * L);
    double* rexp = (double*)malloc(sizeof(double) * (size_t)B * L); /* soft-argmax r per row (unscaled k+1 sum) */
    for (int b = 0; b < B; ++b)
        for (int k = 0; k + 1 < L; ++k) {
            int64_t lab = labels[(size_t)b * L + k + 1];
            if (lab == -100) continue;
            const float* z = logits + ((size_t)b * S + (S - L + k)) * V;
            double mx = z[0];
            for (int v = 1; v < V; ++v) if (z[v] > mx) mx = z[v];
            double sm = 0.0;
            for (int v = 0; v < V; ++v) sm += exp((double)z[v] - mx);
            double l = mx + log(sm);
            lse[(size_t)b * L + k] = l;
            ce_sum += l - (double)z[lab];
            double amx = z[A0];
            for (int v = 1; v < NA; ++v) if (z[A0 + v] > amx) amx = z[A0 + v];
            double asum = 0.0, ew = 0.0;
            for (int v = 0; v < NA; ++v) { double e = exp((double)z[A0 + v] - amx); asum += e; ew += e * (v + 1); }
            rexp[(size_t)b * L + k] = ew / asum; /* in [1,256] */
            if (lab > 2) { double r = ew / asum / 256.0, t = (lab > 31872) ? 0.0 : 1.0; mse_sum += (r - t) * (r - t); }
        }
    const double ce = nrows ? ce_sum / nrows : 0.0;
    const double mse = nact ? w * w * mse_sum / nact : 0.0;
    double total = 0.0, aux0 = 0.0, aux1 = 0.0;
    /* UPA per-sample quantities */
    double *cs = NULL, *nd = NULL; double mean_norm = 0.0;
    if (mode == 2) {
        cs = (double*)calloc(B, sizeof(double));
        nd = (double*)calloc(B, sizeof(double));
        for (int b = 0; b < B; ++b) {
            double e3[3], l3[3]; int cnt = 0;
            for (int k = 0; k + 1 < L && cnt < 3; ++k) {
                int64_t lab = labels[(size_t)b * L + k + 1];
                if (lab == -100) continue;
                e3[cnt] = (rexp[(size_t)b * L + k] - 1.0) / 255.0;
                l3[cnt] = ((double)(lab - 31743) - 1.0) / 255.0;
                ++cnt;
            }
            double dot = 0, ne = 0, nl = 0, d2 = 0;
            for (int q = 0; q < 3; ++q) { dot += e3[q] * l3[q]; ne += e3[q] * e3[q]; nl += l3[q] * l3[q]; d2 += (e3[q] - l3[q]) * (e3[q] - l3[q]); }
            double den = fmax(sqrt(ne), 1e-8) * fmax(sqrt(nl), 1e-8); /* F.cosine_similarity eps=1e-8 [3p torch] */
            cs[b] = dot / den;
            nd[b] = sqrt(d2);
            aux0 += cs[b] + 1.0;
            mean_norm += nd[b];
        }
        aux0 /= B; mean_norm /= B;
        aux1 = 1.0 / (mean_norm + 1e-3);
        total = alpha * aux0 + beta * aux1;
    } else if (mode == 0) total = mse + 1.0 / ce;
    else if (mode == 1) total = mse;
    else total = scale * ce;
    scalars[0] = (float)total; scalars[1] = (float)ce; scalars[2] = (float)mse; scalars[3] = (float)aux0;
    scalars[4] = (float)aux1; scalars[5] = (float)nrows; scalars[6] = (float)nact; scalars[7] = 0.0f;

    if (glog) {
        const double dce = (mode == 0) ? -1.0 / (ce * ce) : (mode == 3 ? scale : 0.0); /* d total / d CE */
        for (int b = 0; b < B; ++b) {
            int cnt = 0;
            double e3[3] = {0, 0, 0}, l3[3] = {0, 0, 0}; int kk[3] = {-1, -1, -1};
            if (mode == 2) {
                for (int k = 0; k + 1 < L && cnt < 3; ++k) {
                    int64_t lab = labels[(size_t)b * L + k + 1];
                    if (lab == -100) continue;
                    e3[cnt] = (rexp[(size_t)b * L + k] - 1.0) / 255.0; l3[cnt] = ((double)(lab - 31743) - 1.0) / 255.0; kk[cnt] = k; ++cnt;
                }
            }
            for (int k = 0; k + 1 < L; ++k) {
                int64_t lab = labels[(size_t)b * L + k + 1];
                if (lab == -100) continue;
                const float* z = logits + ((size_t)b * S + (S - L + k)) * V;
                float* g = glog + ((size_t)b * S + (S - L + k)) * V;
                const double l = lse[(size_t)b * L + k];
                for (int v = 0; v < V; ++v) g[v] = (float)(dce * (exp((double)z[v] - l) - (v == lab ? 1.0 : 0.0)) / nrows);
                /* d total / d E where E = sum p_k (k+1) over the action slice of this row */
                double dE = 0.0;
                if ((mode == 0 || mode == 1) && lab > 2) {
                    double r = rexp[(size_t)b * L + k] / 256.0, t = (lab > 31872) ? 0.0 : 1.0;
                    dE = w * w * 2.0 * (r - t) / nact / 256.0;
                } else if (mode == 2) {
                    for (int q = 0; q < 3; ++q) if (kk[q] == k) {
                        double ne = 0, nl = 0, dot = 0;
                        for (int t = 0; t < 3; ++t) { ne += e3[t] * e3[t]; nl += l3[t] * l3[t]; dot += e3[t] * l3[t]; }
                        double sne = fmax(sqrt(ne), 1e-8), snl = fmax(sqrt(nl), 1e-8);
                        double dcos = l3[q] / (sne * snl) - dot * e3[q] / (sne * sne * sne * snl);
                        double dn = nd[b] > 0 ? (e3[q] - l3[q]) / nd[b] : 0.0;
                        double dtot_de = alpha * dcos / B + beta * (-aux1 * aux1) * dn / B;
                        dE = dtot_de / 255.0;
                    }
                }
                if (dE != 0.0) {
                    double amx = z[A0];
                    for (int v = 1; v < NA; ++v) if (z[A0 + v] > amx) amx = z[A0 + v];
                    double asum = 0.0;
                    for (int v = 0; v < NA; ++v) asum += exp((double)z[A0 + v] - amx);
                    const double E = rexp[(size_t)b * L + k];
                    for (int v = 0; v < NA; ++v) {
                        double p = exp((double)z[A0 + v] - amx) / asum;
                        g[A0 + v] += (float)(dE * p * ((double)(v + 1) - E));
                    }
                }
            }
        }
    }
    free(lse); free(rexp); free(cs); free(nd);
}

/* argmax over the action slice per action row -> predicted token ids (UADA.py:395), in (b,k) order; returns count */
int vaa_oracle_action_argmax(const float* logits, const int64_t* labels, int B, int S, int L, int V, int64_t* pred, int64_t* gt) {
    int n = 0;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k + 1 < L; ++k) {
            int64_t lab = labels[(size_t)b * L + k + 1];
            if (lab <= 2) continue;
            const float* z = logits + ((size_t)b * S + (S - L + k)) * V + 31744;
            int best = 0;
            for (int v = 1; v < 256; ++v) if (z[v] > z[best]) best = v;
            pred[n] = 31744 + best; gt[n] = lab; ++n;
        }
    return n;
}

/* K4: transformers==4.40.1 AdamW.step [3p, parity unpinned] + clamp(0,1) (UADA.py:155-156), or PGD sign step
 * (TMA.py:171-175); optional L1 grad-norm clip (UPA.py:157) and 1/world gradient scale (DDP mean, UADA_ddp.py:166).
 * mode 0 = ADAMW_HF, 1 = PGD_SIGN. step = 1-based Adam step count t. Returns sum|g| (after grad_scale, before clip). */
float vaa_oracle_patch_update(float* patch, const float* g_in, float* m, float* v, int n, int mode, float lr, float b1, float b2,
                              float eps, int step, float l1_clip, float grad_scale) {
    double l1 = 0.0;
    for (int i = 0; i < n; ++i) l1 += fabs((double)(g_in[i] * grad_scale));
    float coef = 1.0f;
    if (l1_clip > 0.0f) { float c = l1_clip / ((float)l1 + 1e-6f); coef = c < 1.0f ? c : 1.0f; }
    const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
    const float step_size = (float)((double)lr * sqrt(1.0 - pow((double)b2, step)) / (1.0 - pow((double)b1, step)));
    (void)bc1; (void)bc2;
    for (int i = 0; i < n; ++i) {
        float g = g_in[i] * grad_scale * coef;
        float p = patch[i];
        if (mode == 0) {
            m[i] = m[i] * b1 + g * (1.0f - b1);
            v[i] = v[i] * b2 + (g * g) * (1.0f - b2);
            float denom = sqrtf(v[i]) + eps;
            p = p - step_size * (m[i] / denom);
        } else {
            float sg = (g > 0.0f) ? 1.0f : (g < 0.0f ? -1.0f : 0.0f);
            p = p - lr * sg;
        }
        patch[i] = fminf(1.0f, fmaxf(0.0f, p));
    }
    return (float)l1;
}

/* Eval-time paste: RandomPatchTransform.simulation_random_patch (appply_random_transform.py:43-78).
 * The patch is quantised like torchvision ToPILImage on a float tensor [3p]: mul(255).byte() (truncation), pasted on the
 * -100 canvas as float, optionally warped (same affine_grid/grid_sample numerics as K1), composited where canvas >= 0,
 * and the float result is truncated to uint8 (numpy astype). img_u8 / out_u8 are [B,224,224,3] HWC. */
void vaa_oracle_patch_apply_eval(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, int B, int ph,
                                 int pw, const int32_t* geometry /*[B]*/, uint8_t* out_u8) {
    float bgrid[IMG];
    base_grid(bgrid);
    const int n = 3 * ph * pw;
    float* q = (float*)malloc(sizeof(float) * n);
    for (int k = 0; k < n; ++k) q[k] = (float)(uint8_t)(patch[k] * 255.0f); /* mul(255).byte() */
    for (int b = 0; b < B; ++b) {
        const int px = xy[2 * b], py = xy[2 * b + 1];
        const float* th = theta + 6 * b;
        for (int i = 0; i < IMG; ++i)
            for (int j = 0; j < IMG; ++j) {
                samp_t s = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f};
                if (geometry[b]) s = sample_pos(bgrid, th, i, j);
                for (int c = 0; c < 3; ++c) {
                    float cv = geometry[b] ? sample_canvas(q, c, ph, pw, px, py, &s) : canvas_at(q, c, ph, pw, px, py, j, i);
                    size_t o = ((size_t)(b * IMG + i) * IMG + j) * 3 + c;
                    out_u8[o] = (cv < 0.0f) ? img_u8[o] : (uint8_t)cv; /* torch.where(canvas < 0, image, canvas) -> astype(uint8) */
                }
            }
    }
    free(q);
}

/* ---------------------------------------------------------------------------------------------
 * resize_patch=True (BASELINE config 5): appply_random_transform.py:113-116 with the Appendix A-D2 semantics
 * (every image scales the BASE patch): `transforms.Resize((h, w))(patch)` [3p torchvision 0.17 -> torch
 * F.interpolate(mode='bilinear', antialias=True, align_corners=False)]. The restatement below follows torch's CPU kernel
 * (ATen UpSampleKernel.cpp: _compute_indices_min_size_weights_aa + separable horizontal-then-vertical passes, each
 * output = src[0]*w[0] then fma(src[j], w[j], acc)); tests/test_oracle_golden.py checks it BIT-EXACT against
 * F.interpolate for a sweep of sizes.
 * pdesc [B,4] int32 = {h_b, w_b, offset_b (in floats into `packed`), 0}; packed holds [3,h_b,w_b] per image.
 * --------------------------------------------------------------------------------------------- */
static int aa_weights(int i, int in_size, float scale, float support, int max_interp, float* wt, int* xmin_o) {
    float center = (float)((double)scale * ((double)i + 0.5));
    float total_w = 0.0f;
    float invscale = (scale >= 1.0f) ? (float)(1.0 / (double)scale) : 1.0f;
    int64_t xmin = (int64_t)((double)(center - support) + 0.5);
    if (xmin < 0) xmin = 0;
    int64_t xe = (int64_t)((double)(center + support) + 0.5);
    if (xe > in_size) xe = in_size;
    int64_t xsize = xe - xmin;
    if (xsize < 0) xsize = 0;
    if (xsize > max_interp) xsize = max_interp;
    for (int j = 0; j < xsize; ++j) {
        float x = (float)(((double)((float)(j + xmin) - center) + 0.5) * (double)invscale);
        x = fabsf(x);
        float w = (x < 1.0f) ? (float)(1.0 - (double)x) : 0.0f; /* HelperInterpLinear::aa_filter */
        wt[j] = w;
        total_w += w;
    }
    if (total_w != 0.0f)
        for (int j = 0; j < xsize; ++j) wt[j] /= total_w;
    *xmin_o = (int)xmin;
    return (int)xsize;
}

static void aa_axis(int in_size, int out_size, float* scale, float* support, int* max_interp) {
    *scale = (float)in_size / (float)out_size; /* area_pixel_compute_scale, align_corners=False, no scale_factor */
    *support = (*scale >= 1.0f) ? *scale : 1.0f;
    *max_interp = (int)ceilf(*support) * 2 + 1;
}

static void resize_plane_fwd(const float* src, int ih, int iw, float* dst, int oh, int ow) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)ih * ow);
    const float* hsrc = src;
    if (ow != iw) { /* horizontal pass (skipped when the width is unchanged, as torch does) */
        float scale, support;
        int mi;
        aa_axis(iw, ow, &scale, &support, &mi);
        float* wt = (float*)malloc(sizeof(float) * mi);
        for (int x = 0; x < ow; ++x) {
            int xmin, n = aa_weights(x, iw, scale, support, mi, wt, &xmin);
            for (int y = 0; y < ih; ++y) {
                const float* s = src + (size_t)y * iw + xmin;
                float o = s[0] * wt[0];
                for (int j = 1; j < n; ++j) o = fmaf(s[j], wt[j], o);
                tmp[(size_t)y * ow + x] = o;
            }
        }
        free(wt);
        hsrc = tmp;
    }
    if (oh != ih) { /* vertical pass */
        float scale, support;
        int mi;
        aa_axis(ih, oh, &scale, &support, &mi);
        float* wt = (float*)malloc(sizeof(float) * mi);
        for (int y = 0; y < oh; ++y) {
            int ymin, n = aa_weights(y, ih, scale, support, mi, wt, &ymin);
            for (int x = 0; x < ow; ++x) {
                const float* s = hsrc + (size_t)ymin * ow + x;
                float o = s[0] * wt[0];
                for (int j = 1; j < n; ++j) o = fmaf(s[(size_t)j * ow], wt[j], o);
                dst[(size_t)y * ow + x] = o;
            }
        }
        free(wt);
    } else {
        memcpy(dst, hsrc, sizeof(float) * (size_t)oh * ow);
    }
    free(tmp);
}

void vaa_oracle_patch_resize_fwd(const float* patch, int ph, int pw, const int32_t* pdesc, int B, float* packed) {
    for (int b = 0; b < B; ++b) {
        const int h = pdesc[4 * b], w = pdesc[4 * b + 1];
        float* dst = packed + pdesc[4 * b + 2];
        for (int c = 0; c < 3; ++c) resize_plane_fwd(patch + (size_t)c * ph * pw, ph, pw, dst + (size_t)c * h * w, h, w);
    }
}

/* autograd backward of the resize to the base patch (torch cpu_upsample_genNd_backward_aa: grad_in[ymin+y][xmin+x] +=
 * wx[x]*wy[y]*g in (oh, ow) scan order), summed over the images last-created-first like autograd does. */
void vaa_oracle_patch_resize_bwd(const float* gpacked, int ph, int pw, const int32_t* pdesc, int B, float* gpatch) {
    const size_t n = (size_t)3 * ph * pw;
    float* acc = (float*)malloc(sizeof(float) * n);
    memset(gpatch, 0, sizeof(float) * n);
    for (int b = B - 1; b >= 0; --b) {
        const int h = pdesc[4 * b], w = pdesc[4 * b + 1];
        const float* g = gpacked + pdesc[4 * b + 2];
        memset(acc, 0, sizeof(float) * n);
        float sh, sw, suph, supw;
        int mih, miw;
        aa_axis(ph, h, &sh, &suph, &mih);
        aa_axis(pw, w, &sw, &supw, &miw);
        float* wy = (float*)malloc(sizeof(float) * mih);
        float* wx = (float*)malloc(sizeof(float) * miw);
        for (int oh = 0; oh < h; ++oh) {
            int ymin, ysize = aa_weights(oh, ph, sh, suph, mih, wy, &ymin);
            for (int ow = 0; ow < w; ++ow) {
                int xmin, xsize = aa_weights(ow, pw, sw, supw, miw, wx, &xmin);
                for (int c = 0; c < 3; ++c) {
                    const float gv = g[((size_t)c * h + oh) * w + ow];
                    for (int y = 0; y < ysize; ++y)
                        for (int x = 0; x < xsize; ++x) acc[((size_t)c * ph + ymin + y) * pw + xmin + x] += wx[x] * wy[y] * gv;
                }
            }
        }
        free(wy);
        free(wx);
        for (size_t k = 0; k < n; ++k) gpatch[k] += acc[k];
    }
    free(acc);
}

/* K1 / K2 with one patch PER IMAGE (the resized patches of config 5): image b uses packed + offset_b as [3,h_b,w_b]. */
void vaa_oracle_patch_apply_fwd_multi(const uint8_t* img_u8, const float* packed, const int32_t* pdesc, const int32_t* xy,
                                      const float* theta, int B, int geometry, int mask_mode, const float* mean6, const float* std6,
                                      float* out_f32, uint16_t* out_bf16, uint8_t* keep) {
    for (int b = 0; b < B; ++b)
        vaa_oracle_patch_apply_fwd(img_u8 + (size_t)b * NPIX * 3, packed + pdesc[4 * b + 2], xy + 2 * b, theta + 6 * b, 1, pdesc[4 * b],
                                   pdesc[4 * b + 1], geometry, mask_mode, mean6, std6, out_f32 ? out_f32 + (size_t)b * 6 * NPIX : NULL,
                                   out_bf16 ? out_bf16 + (size_t)b * 6 * NPIX : NULL, keep ? keep + (size_t)b * 3 * NPIX : NULL);
}

/* gpacked: d L / d packed (every image's own resized patch), same layout as packed; regions between patches untouched. */
void vaa_oracle_patch_grad_multi(const uint16_t* gout_bf16, const float* packed, const int32_t* pdesc, const int32_t* xy,
                                 const float* theta, int B, int geometry, int mask_mode, const float* std6, float* gpacked) {
    for (int b = 0; b < B; ++b)
        vaa_oracle_patch_grad(gout_bf16 + (size_t)b * 6 * NPIX, packed + pdesc[4 * b + 2], xy + 2 * b, theta + 6 * b, 1, pdesc[4 * b],
                              pdesc[4 * b + 1], geometry, mask_mode, std6, gpacked + pdesc[4 * b + 2]);
}
