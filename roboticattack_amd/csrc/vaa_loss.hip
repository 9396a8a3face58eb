// vaa_loss.hip — K3: action-token discrepancy losses, forward + gradient w.r.t. the logits, labelled rows only.
//
// Replaces HF Llama's `.loss` (mean CE over shifted non-ignored labels, all 32064 classes; reached through
// modeling_prismatic.py:404-415) plus OpenVLAAttacker.weighted_loss (UADA.py:381-406, UADA_ddp.py:99-124,
// UPA.py:367-387) and their autograd backward. The reference materialises fp32 logits [B,S,32064] and runs
// softmax/CE over every one of the B*S rows; only B*(n_mask+1) rows carry loss, so these kernels touch
// exactly those rows: one streaming read for the statistics, one read + one write for the gradient.
//
//   stats    : grid = B*(L-1) positions, unlabelled positions exit at once; 1024 threads stream one row
//              (16-byte loads), block-reduce max / sum-exp, wave 0 does the 256-wide action slice.
//   finalize : one workgroup folds the per-row statistics into the scalars and per-row gradient coefficients
//              (needs global means: 1/CE, UPA's mean norm) in fixed order -> deterministic.
//   grad     : same grid as stats; g = kCE*(softmax - onehot) + kE*p_a*((a+1) - E) on the action slice.
#include "vaa_common.h"

namespace vaa {

constexpr int kA0 = 31744;   // first action token (UADA.py:384)
constexpr int kNA = 256;     // action bins
constexpr int kRowThreads = 1024;

struct RowStat {  // per position p = b*(L-1)+k
    float lse;      // logsumexp over all V classes
    float zlab;     // logit of the label
    float alse;     // logsumexp over the 256 action classes
    float E;        // sum_a softmax_a * (a+1), in [1,256]
    int pred;       // 31744 + argmax over the action slice
    int rowidx;     // compact row index (ROWS layout) or -1
    float kce;      // d total / d z contribution weight of (softmax - onehot)
    float kE;       // d total / d E
};

struct LossArgs {
    const void* logits;
    const int64_t* labels;
    RowStat* st;
    void* glogits;
    float* scalars;
    int32_t* pred_tokens;
    int B, S, L, V, mode, layout;
    float w, alpha, beta, scale;
};

__device__ __forceinline__ size_t row_offset(const LossArgs& a, int b, int k, int rowidx) {
    return a.layout == VAA_LAYOUT_FULL ? ((size_t)b * a.S + (a.S - a.L + k)) * a.V : (size_t)rowidx * a.V;
}

template <typename T>
struct Vec;
template <>
struct Vec<float> {
    static constexpr int N = 4;
    typedef float4 raw;
    __device__ static void load(const float* p, float* v) {
        float4 r = *reinterpret_cast<const float4*>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    __device__ static void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    __device__ static float get(const float* p) { return *p; }
    __device__ static void put(float* p, float v) { *p = v; }
};
template <>
struct Vec<uint16_t> {  // bf16 bits
    static constexpr int N = 8;
    __device__ static void load(const uint16_t* p, float* v) {
        uint4 r = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
    __device__ static void store(uint16_t* p, const float* v) {
        uint4 r;
        r.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        r.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        r.z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
        r.w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
        *reinterpret_cast<uint4*>(p) = r;
    }
    __device__ static float get(const uint16_t* p) { return bf16_bits_to_f32(*p); }
    __device__ static void put(uint16_t* p, float v) { *p = (uint16_t)f32_to_bf16_bits(v); }
};

// ---- index: compact row numbers in (b,k) row-major order of labelled positions (ROWS layout) ----
__global__ __launch_bounds__(1024) void loss_index_kernel(const int64_t* __restrict__ labels, RowStat* st, int B, int L) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        int cnt = 0;
        if (b < B)
            for (int k = 0; k + 1 < L; ++k) cnt += labels[(size_t)b * L + k + 1] != -100;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int base = carry;
        for (int q = 0; q < wv; ++q) base += wsum[q];
        int r = base + incl - cnt;
        if (b < B)
            for (int k = 0; k + 1 < L; ++k) {
                const bool lab = labels[(size_t)b * L + k + 1] != -100;
                st[(size_t)b * (L - 1) + k].rowidx = lab ? r : -1;
                r += lab;
            }
        __syncthreads();
        if (tid == 1023) carry = base + incl;
        __syncthreads();
    }
}

// ---- stats ----
template <typename T>
__global__ __launch_bounds__(kRowThreads) void loss_stats_kernel(LossArgs a) {
    const int p = blockIdx.x, b = p / (a.L - 1), k = p - b * (a.L - 1);
    const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
    if (lab == -100) return;
    RowStat* st = a.st + p;
    const T* z = reinterpret_cast<const T*>(a.logits) + row_offset(a, b, k, st->rowidx);
    constexpr int N = Vec<T>::N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nvec = a.V / N;  // V = 32064 is a multiple of 8
    __shared__ float red[16];
    __shared__ float bmax;

    // one streaming pass: this thread's elements stay in registers (8 x 16-byte vectors cover V = 32064 in f32)
    constexpr int MAXV = 32 / N;  // 8 f32 or 4 bf16 vectors per thread = 32 logits x 1024 threads >= 32064
    float v[MAXV][N];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = tid + c * kRowThreads;
        if (q < nvec) {
            Vec<T>::load(z + (size_t)q * N, v[c]);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) v[c][e] = -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < N; ++e) m = fmaxf(m, v[c][e]);
    }
    // generic tail for V larger than MAXV*N*1024 (not the case for OpenVLA): re-read from L2
    for (int q = tid + MAXV * kRowThreads; q < nvec; q += kRowThreads) {
        float t[N];
        Vec<T>::load(z + (size_t)q * N, t);
#pragma unroll
        for (int e = 0; e < N; ++e) m = fmaxf(m, t[e]);
    }
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    if (tid == 0) {
        float mm = red[0];
        for (int q = 1; q < kRowThreads / 64; ++q) mm = fmaxf(mm, red[q]);
        bmax = mm;
    }
    __syncthreads();
    const float M = bmax;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c)
#pragma unroll
        for (int e = 0; e < N; ++e) s += expf(v[c][e] - M);  // exp(-inf) = 0 for the padding
    for (int q = tid + MAXV * kRowThreads; q < nvec; q += kRowThreads) {
        float t[N];
        Vec<T>::load(z + (size_t)q * N, t);
#pragma unroll
        for (int e = 0; e < N; ++e) s += expf(t[e] - M);
    }
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wv] = s;
    __syncthreads();

    if (wv == 0) {  // action slice: 256 classes, 4 per lane
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = Vec<T>::get(z + kA0 + lane * 4 + e);
        float am = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        int ai = 0;
#pragma unroll
        for (int e = 1; e < 4; ++e) if (x[e] > x[ai]) ai = e;
        float bestv = x[ai];
        int besti = lane * 4 + ai;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {  // argmax with lowest-index tie break (torch.argmax on CPU)
            float ov = __shfl_xor(bestv, o, 64);
            int oi = __shfl_xor(besti, o, 64);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        am = wave_max(am);
        float es = 0.0f, ew = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ex = expf(x[e] - am);
            es += ex;
            ew += ex * (float)(lane * 4 + e + 1);
        }
        es = wave_sum(es);
        ew = wave_sum(ew);
        if (lane == 0) {
            float tot = 0.0f;
            for (int q = 0; q < kRowThreads / 64; ++q) tot += red[q];
            st->lse = M + logf(tot);
            st->zlab = Vec<T>::get(z + lab);
            st->alse = am + logf(es);
            st->E = ew / es;
            st->pred = kA0 + besti;
        }
    }
}

// ---- finalize: scalars + per-row gradient coefficients (single workgroup, fixed-order reductions) ----
__device__ double block_sum(double v, double* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < 16; ++q) t += sh[q];
    return t;
}

__device__ __forceinline__ double bin_center(int tok) {  // ActionTokenizer.decode_token_ids_to_actions (action_tokenizer.py:49-68)
    int d = 32000 - tok - 1;
    d = d < 0 ? 0 : (d > 254 ? 254 : d);
    return -1.0 + (2.0 * d + 1.0) / 255.0;
}

__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossArgs a) {
    __shared__ double sh[16];
    const int tid = threadIdx.x;
    const int P = a.B * (a.L - 1);
    double ce = 0.0, mse = 0.0, uad = 0.0, nrow = 0.0, nact = 0.0;
    for (int p = tid; p < P; p += 1024) {
        const int b = p / (a.L - 1), k = p - b * (a.L - 1);
        const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
        if (a.pred_tokens) a.pred_tokens[p] = (lab > 2) ? a.st[p].pred : -1;
        if (lab == -100) continue;
        const RowStat s = a.st[p];
        nrow += 1.0;
        ce += (double)s.lse - (double)s.zlab;
        if (lab > 2) {
            nact += 1.0;
            const double r = (double)s.E / 256.0, t = (lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10: 1/256 -> 0)
            mse += (r - t) * (r - t);
            const double ag = bin_center((int)lab), ap = bin_center(s.pred);  // cal_UAD, UADA.py:408-418
            uad += fabs(ap - ag) / (ag > 0 ? fabs(ag + 1.0) : fabs(ag - 1.0));
        }
    }
    ce = block_sum(ce, sh); mse = block_sum(mse, sh); uad = block_sum(uad, sh);
    nrow = block_sum(nrow, sh); nact = block_sum(nact, sh);
    const double CE = nrow > 0 ? ce / nrow : 0.0;
    const double MSE = nact > 0 ? (double)a.w * a.w * mse / nact : 0.0;
    const double UAD = nact > 0 ? uad / nact : 0.0;

    double total = 0.0, aux0 = 0.0, aux1 = 0.0;
    if (a.mode == VAA_LOSS_UPA) {
        // per sample: first three labelled positions = x,y,z tokens (UPA.py:375-380)
        double ang = 0.0, nsum = 0.0;
        for (int b = tid; b < a.B; b += 1024) {
            double e3[3] = {0, 0, 0}, l3[3] = {0, 0, 0};
            int cnt = 0;
            for (int k = 0; k + 1 < a.L && cnt < 3; ++k) {
                const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
                if (lab == -100) continue;
                e3[cnt] = ((double)a.st[(size_t)b * (a.L - 1) + k].E - 1.0) / 255.0;
                l3[cnt] = ((double)(lab - 31743) - 1.0) / 255.0;
                ++cnt;
            }
            double dot = 0, ne = 0, nl = 0, d2 = 0;
            for (int q = 0; q < 3; ++q) { dot += e3[q] * l3[q]; ne += e3[q] * e3[q]; nl += l3[q] * l3[q]; d2 += (e3[q] - l3[q]) * (e3[q] - l3[q]); }
            ang += dot / (fmax(sqrt(ne), 1e-8) * fmax(sqrt(nl), 1e-8)) + 1.0;  // F.cosine_similarity + 1 (UPA.py:382-383)
            nsum += sqrt(d2);
        }
        ang = block_sum(ang, sh);
        nsum = block_sum(nsum, sh);
        aux0 = ang / a.B;
        const double mean_norm = nsum / a.B;
        aux1 = 1.0 / (mean_norm + 1e-3);  // UPA.py:384
        total = (double)a.alpha * aux0 + (double)a.beta * aux1;
        for (int b = tid; b < a.B; b += 1024) {
            double e3[3] = {0, 0, 0}, l3[3] = {0, 0, 0};
            int kk[3] = {-1, -1, -1}, cnt = 0;
            for (int k = 0; k + 1 < a.L; ++k) {
                const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
                if (lab == -100) continue;
                RowStat* s = a.st + (size_t)b * (a.L - 1) + k;
                s->kce = 0.0f;
                s->kE = 0.0f;
                if (cnt < 3) { e3[cnt] = ((double)s->E - 1.0) / 255.0; l3[cnt] = ((double)(lab - 31743) - 1.0) / 255.0; kk[cnt] = k; ++cnt; }
            }
            double dot = 0, ne = 0, nl = 0, d2 = 0;
            for (int q = 0; q < 3; ++q) { dot += e3[q] * l3[q]; ne += e3[q] * e3[q]; nl += l3[q] * l3[q]; d2 += (e3[q] - l3[q]) * (e3[q] - l3[q]); }
            const double sne = fmax(sqrt(ne), 1e-8), snl = fmax(sqrt(nl), 1e-8), nd = sqrt(d2);
            for (int q = 0; q < cnt; ++q) {
                const double dcos = l3[q] / (sne * snl) - dot * e3[q] / (sne * sne * sne * snl);
                const double dn = nd > 0 ? (e3[q] - l3[q]) / nd : 0.0;
                const double dde = (double)a.alpha * dcos / a.B - (double)a.beta * aux1 * aux1 * dn / a.B;
                a.st[(size_t)b * (a.L - 1) + kk[q]].kE = (float)(dde / 255.0);
            }
        }
    } else {
        double dce = 0.0;
        if (a.mode == VAA_LOSS_UADA) { total = MSE + 1.0 / CE; dce = -1.0 / (CE * CE); }  // UADA.py:147
        else if (a.mode == VAA_LOSS_UADA_DDP) { total = MSE; }                            // UADA_ddp.py:203-206
        else { total = (double)a.scale * CE; dce = (double)a.scale; }                     // TMA.py:148
        const float kce = nrow > 0 ? (float)(dce / nrow) : 0.0f;
        const bool has_mse = a.mode != VAA_LOSS_CE;
        for (int p = tid; p < P; p += 1024) {
            const int b = p / (a.L - 1), k = p - b * (a.L - 1);
            const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
            if (lab == -100) continue;
            RowStat* s = a.st + p;
            s->kce = kce;
            float kE = 0.0f;
            if (has_mse && lab > 2) {
                const double r = (double)s->E / 256.0, t = (lab > 31872) ? 0.0 : 1.0;
                kE = (float)((double)a.w * a.w * 2.0 * (r - t) / nact / 256.0);
            }
            s->kE = kE;
        }
    }
    if (tid == 0) {
        a.scalars[0] = (float)total; a.scalars[1] = (float)CE; a.scalars[2] = (float)MSE; a.scalars[3] = (float)aux0;
        a.scalars[4] = (float)aux1; a.scalars[5] = (float)nrow; a.scalars[6] = (float)nact; a.scalars[7] = (float)UAD;
    }
}

// ---- grad ----
template <typename T>
__global__ __launch_bounds__(kRowThreads) void loss_grad_kernel(LossArgs a) {
    const int p = blockIdx.x, b = p / (a.L - 1), k = p - b * (a.L - 1);
    const int64_t lab = a.labels[(size_t)b * a.L + k + 1];
    if (lab == -100) return;
    const RowStat s = a.st[p];
    const size_t off = row_offset(a, b, k, s.rowidx);
    const T* z = reinterpret_cast<const T*>(a.logits) + off;
    T* g = reinterpret_cast<T*>(a.glogits) + off;
    constexpr int N = Vec<T>::N;
    const int nvec = a.V / N;
    for (int q = threadIdx.x; q < nvec; q += kRowThreads) {
        float x[N], o[N];
        const int v0 = q * N;
        const bool in_slice = (v0 >= kA0 && v0 < kA0 + kNA);
        if (s.kce != 0.0f || (in_slice && s.kE != 0.0f)) Vec<T>::load(z + (size_t)v0, x);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float gv = 0.0f;
            if (s.kce != 0.0f) gv = s.kce * (expf(x[e] - s.lse) - ((v0 + e) == lab ? 1.0f : 0.0f));
            if (in_slice && s.kE != 0.0f) {
                const float pa = expf(x[e] - s.alse);
                gv += s.kE * pa * ((float)(v0 + e - kA0 + 1) - s.E);
            }
            o[e] = gv;
        }
        Vec<T>::store(g + (size_t)v0, o);
    }
}

}  // namespace vaa

extern "C" size_t vaa_loss_ws_bytes(int B, int L) {
    if (B <= 0 || L <= 1) return 0;
    return (size_t)B * (size_t)(L - 1) * sizeof(vaa::RowStat);
}

extern "C" int vaa_loss_fwd_bwd(const void* logits, int dtype, int layout, const int64_t* labels, int B, int S, int L, int V,
                                int mode, const float* params, float* scalars, int32_t* pred_tokens, void* glogits, void* ws,
                                size_t ws_bytes, void* stream) {
    using namespace vaa;
    if (!logits || !labels || !params || !scalars) {
        set_error("vaa_loss_fwd_bwd: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B <= 0 || L <= 1 || V < kA0 + kNA || (V % 8) != 0 || mode < 0 || mode > VAA_LOSS_CE ||
        (dtype != VAA_DTYPE_F32 && dtype != VAA_DTYPE_BF16) || (layout != VAA_LAYOUT_FULL && layout != VAA_LAYOUT_ROWS)) {
        set_error("vaa_loss_fwd_bwd: bad sizes/mode (B=%d L=%d V=%d mode=%d dtype=%d layout=%d)", B, L, V, mode, dtype, layout);
        return VAA_E_INVALID;
    }
    if (layout == VAA_LAYOUT_FULL && S < L) {
        set_error("vaa_loss_fwd_bwd: FULL layout needs S >= L (S=%d L=%d)", S, L);
        return VAA_E_INVALID;
    }
    if (!ws || ws_bytes < vaa_loss_ws_bytes(B, L)) {
        set_error("vaa_loss_fwd_bwd: workspace %zu B < required %zu B", ws_bytes, vaa_loss_ws_bytes(B, L));
        return VAA_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    LossArgs a;
    a.logits = logits; a.labels = labels; a.st = (RowStat*)ws; a.glogits = glogits; a.scalars = scalars; a.pred_tokens = pred_tokens;
    a.B = B; a.S = S; a.L = L; a.V = V; a.mode = mode; a.layout = layout;
    a.w = params[0]; a.alpha = params[1]; a.beta = params[2]; a.scale = params[3];
    const int P = B * (L - 1);
    if (layout == VAA_LAYOUT_ROWS) {
        hipLaunchKernelGGL(loss_index_kernel, dim3(1), dim3(1024), 0, st, labels, a.st, B, L);
        int rc = check_launch("vaa_loss_fwd_bwd(index)");
        if (rc != VAA_OK) return rc;
    }
    if (dtype == VAA_DTYPE_F32) hipLaunchKernelGGL(loss_stats_kernel<float>, dim3(P), dim3(kRowThreads), 0, st, a);
    else hipLaunchKernelGGL(loss_stats_kernel<uint16_t>, dim3(P), dim3(kRowThreads), 0, st, a);
    int rc = check_launch("vaa_loss_fwd_bwd(stats)");
    if (rc != VAA_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, a);
    rc = check_launch("vaa_loss_fwd_bwd(finalize)");
    if (rc != VAA_OK) return rc;
    if (glogits) {
        if (dtype == VAA_DTYPE_F32) hipLaunchKernelGGL(loss_grad_kernel<float>, dim3(P), dim3(kRowThreads), 0, st, a);
        else hipLaunchKernelGGL(loss_grad_kernel<uint16_t>, dim3(P), dim3(kRowThreads), 0, st, a);
        rc = check_launch("vaa_loss_fwd_bwd(grad)");
    }
    return rc;
}
