// vaa_update.hip — K4: fused pixel update of the patch: [DDP 1/world scale] -> [L1 grad-norm clip] ->
// HF-AdamW or PGD-sign step -> clamp to [0,1], plus the logged gradient statistics. The patch has 7,500 (50x50) to 30,000 (100x100; any size
// up to the frame is accepted) elements: the op is latency bound, and the L1 norm / the statistics need the whole gradient first — every
// workgroup computes them redundantly (bitwise the same), then updates its share of the elements.
//
// Replaces (UADA.py:154-157, UADA_ddp.py:207-209): `patch.grad.mean().item()`, `optimizer.step()`
// (transformers==4.40.1 AdamW [3p]: m=b1*m+(1-b1)g; v=b2*v+(1-b2)g^2; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps),
// eps NOT bias corrected), `patch.data.clamp(0,1)`, `zero_grad()`; UPA.py:157 clip_grad_norm_(L1, 1e-3);
// TMA.py:171-175 PGD sign step.
#include <math.h>

#include "vaa_common.h"

namespace vaa {

constexpr int kUpdRegsSmall = 8;  // statistics pass: gradient elements a thread holds — patches up to 8,192 elements (3x50x50 = 7,500) ...
constexpr int kUpdRegsMid = 32;   // ... and up to 32,768 elements (3x100x100 = 30,000, UPA's resize_patch base patch): ONE memory round trip
constexpr int kUpdPer = 2;        // elements a thread UPDATES: the update is spread over R / kUpdPer workgroups (4 at 50x50, 16 at 100x100)

// The statistics of the logged gradient / the L1 clip need the WHOLE gradient before any element can be updated, and the patch is small
// (30 KB .. 120 KB of gradient): every workgroup computes the statistics of the whole gradient itself — the same loads (L2 hits after the
// first workgroup), the same per-thread order, the same tree, hence the same bits in every workgroup, no grid-wide hand-over — and then
// updates ITS rows of the [R][1024] element layout: row e of workgroup b is e = b + j * G (j < PER). One workgroup doing all of it was
// bound by ONE CU's issue rate and L2 port: sqrt + divide + clamp on 32 elements per thread and 840 KB through one CU took 25-30 us at
// 3x100x100 (the UPA step with resize_patch), 5.x us at 3x50x50.
//   R = 0: any n, one workgroup, streaming (patches beyond 32,768 elements).
// All forms add a thread's elements in the same increasing-index order, so statistics, clip coefficient and updates agree bitwise.
template <int R>
__global__ __launch_bounds__(1024) void patch_update_kernel(UpdArgs a) {
    __shared__ double sh_abs[16], sh_sum[16];
    __shared__ float coef_sh;
    constexpr int RR = R > 0 ? R : 1, PER = R > 0 ? kUpdPer : 1, G = R > 0 ? R / kUpdPer : 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bid = blockIdx.x;
    const bool adam = a.mode == VAA_OPT_ADAMW_HF;
    float gq[RR], g2[PER], p2[PER], m2[PER], v2[PER];
    double sa = 0.0, ss = 0.0;
    if (R > 0) {
        // this workgroup's own elements first (nothing of them depends on the statistics), then the whole gradient: all loads in flight together
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + (bid + j * G) * 1024;
            g2[j] = 0.0f; p2[j] = 0.0f; m2[j] = 0.0f; v2[j] = 0.0f;
            if (i < a.n) {
                g2[j] = a.g[i]; p2[j] = a.patch[i];
                if (adam) { m2[j] = a.m[i]; v2[j] = a.v[i]; }
            }
        }
#pragma unroll
        for (int e = 0; e < RR; ++e) {
            const int i = tid + e * 1024;
            gq[e] = 0.0f;
            if (i < a.n) gq[e] = a.g[i];
        }
#pragma unroll
        for (int e = 0; e < RR; ++e) {  // the same increasing-index order per thread as the streaming form below
            if (tid + e * 1024 < a.n) {
                const float g = gq[e] * a.grad_scale;
                sa += fabs((double)g);
                ss += (double)g;
            }
        }
    } else {
        for (int i = tid; i < a.n; i += 1024) {
            const float g = a.g[i] * a.grad_scale;
            sa += fabs((double)g);
            ss += (double)g;
        }
    }
    sa = wave_sum(sa);
    ss = wave_sum(ss);
    if (lane == 0) { sh_abs[wv] = sa; sh_sum[wv] = ss; }
    __syncthreads();
    if (tid == 0) {
        double ta = 0.0, ts = 0.0;
        for (int q = 0; q < 16; ++q) { ta += sh_abs[q]; ts += sh_sum[q]; }
        float coef = 1.0f;
        if (a.l1_clip > 0.0f) {  // torch clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)
            const float c = a.l1_clip / ((float)ta + 1e-6f);
            coef = c < 1.0f ? c : 1.0f;
        }
        coef_sh = coef;
        if (a.stats && bid == 0) { a.stats[0] = (float)ta; a.stats[1] = (float)(ts / a.n); }
    }
    __syncthreads();
    const float coef = coef_sh;
    if (R > 0) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + (bid + j * G) * 1024;
            if (i < a.n) {
                const float g = g2[j] * a.grad_scale * coef;
                float m = m2[j], v = v2[j];
                const float p = update_one(a, g, p2[j], m, v);
                if (adam) { a.m[i] = m; a.v[i] = v; }
                a.patch[i] = p;
            }
        }
    } else {
        for (int i = tid; i < a.n; i += 1024) {
            const float g = a.g[i] * a.grad_scale * coef;
            float m = adam ? a.m[i] : 0.0f, v = adam ? a.v[i] : 0.0f;
            const float p = update_one(a, g, a.patch[i], m, v);
            if (adam) { a.m[i] = m; a.v[i] = v; }
            a.patch[i] = p;
        }
    }
}

}  // namespace vaa

extern "C" int vaa_patch_update(float* patch, const float* g, float* m, float* v, int n, int mode, float lr, float beta1,
                                float beta2, float eps, int step, float l1_clip, float grad_scale, float* stats, void* stream) {
    using namespace vaa;
    if (!patch || !g || (mode == VAA_OPT_ADAMW_HF && (!m || !v))) {
        set_error("vaa_patch_update: null pointer argument");
        return VAA_E_INVALID;
    }
    if (n <= 0 || (mode != VAA_OPT_ADAMW_HF && mode != VAA_OPT_PGD_SIGN) || (mode == VAA_OPT_ADAMW_HF && step < 1)) {
        set_error("vaa_patch_update: bad sizes/mode (n=%d mode=%d step=%d)", n, mode, step);
        return VAA_E_INVALID;
    }
    {   // several workgroups read the WHOLE gradient while others already write their elements of patch / m / v: no byte of g may lie inside
        // one of those buffers (a view into a shared flat buffer overlaps without being pointer-equal)
        const uintptr_t g0 = (uintptr_t)g, g1 = g0 + (size_t)n * sizeof(float);
        auto overlaps = [&](const float* p) { const uintptr_t p0 = (uintptr_t)p; return p && g0 < p0 + (size_t)n * sizeof(float) && p0 < g1; };
        if (overlaps(patch) || (mode == VAA_OPT_ADAMW_HF && (overlaps(m) || overlaps(v)))) {
            set_error("vaa_patch_update: the gradient must not overlap the patch or a moment buffer");
            return VAA_E_INVALID;
        }
    }
    UpdArgs a;
    a.patch = patch; a.g = g; a.m = m; a.v = v; a.stats = stats; a.n = n; a.mode = mode;
    a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.l1_clip = l1_clip; a.grad_scale = grad_scale;
    // python-side doubles of the reference optimiser, narrowed to f32 exactly where torch narrows them
    const double b1 = (double)beta1, b2 = (double)beta2;
    a.one_m_b1 = (float)(1.0 - b1);
    a.one_m_b2 = (float)(1.0 - b2);
    a.step_size = (mode == VAA_OPT_ADAMW_HF) ? (float)((double)lr * sqrt(1.0 - pow(b2, (double)step)) / (1.0 - pow(b1, (double)step))) : 0.0f;
    // (the gradient must not alias the patch or the moments: a workgroup reads the whole gradient while others already write their rows)
    if (n <= 1024 * kUpdRegsSmall) VAA_LAUNCH(patch_update_kernel<kUpdRegsSmall>, dim3(kUpdRegsSmall / kUpdPer), dim3(1024), 0, (hipStream_t)stream, a);
    else if (n <= 1024 * kUpdRegsMid) VAA_LAUNCH(patch_update_kernel<kUpdRegsMid>, dim3(kUpdRegsMid / kUpdPer), dim3(1024), 0, (hipStream_t)stream, a);
    else VAA_LAUNCH(patch_update_kernel<0>, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    return check_launch("vaa_patch_update");
}
