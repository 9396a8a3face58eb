// vaa_loss.hip — K3: action-token discrepancy losses, forward + gradient w.r.t. the logits, labelled rows only.
//
// Replaces HF Llama's `.loss` (mean CE over shifted non-ignored labels, all 32064 classes; reached through
// modeling_prismatic.py:404-415) plus OpenVLAAttacker.weighted_loss (UADA.py:381-406, UADA_ddp.py:99-124,
// UPA.py:367-387) and their autograd backward. The reference materialises fp32 logits [B,S,32064] and runs
// softmax/CE over every one of the B*S rows; only B*(n_mask+1) rows carry loss, so these kernels touch
// exactly those rows: one streaming read for the statistics, one read + one write for the gradient.
//
//   stats : grid = B*(L-1) positions, unlabelled positions exit at once; 1024 threads stream one row (16-byte loads) into
//           registers, block-reduce max / sum-exp; the 256 action logits sit in one wave's registers -> soft-argmax by shuffles.
//   grad  : same grid; every labelled row folds the per-row statistics into the global scalars it needs (1/CE, UPA means;
//           fixed order -> deterministic), then writes g = kCE*(softmax - onehot) + kE*p_a*((a+1) - E).
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "vaa_common.h"
#include "vaa_rows.h"
#include "vaa_rows_fold.h"

namespace vaa {

constexpr int kRowThreads = 1024;


struct LossArgs {
    const void* logits;
    const int64_t* labels;
    RowStat* st;
    void* glogits;
    float* scalars;
    int32_t* pred_tokens;
    int32_t* pred_full;
    int B, S, L, V, mode, layout;
    float w, alpha, beta, scale;
};

constexpr int kLabLds = 24576;  // label matrices up to this many entries (48 KB as int16) are staged in LDS once per workgroup

__device__ __forceinline__ size_t row_offset(const LossArgs& a, int b, int k, int rowidx) {
    return a.layout == VAA_LAYOUT_FULL ? ((size_t)b * a.S + (a.S - a.L + k)) * a.V : (size_t)rowidx * a.V;
}

template <typename T>
struct Vec;
template <>
struct Vec<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float* v) {
        float4 r = *reinterpret_cast<const float4*>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    __device__ static void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    __device__ static float get(const float* p) { return *p; }
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
};



// Label access: the whole [B,L] matrix staged in LDS as int16 (token ids < 32768, -100 stays -100) when it fits, else global.
struct Labels {
    const int64_t* g;
    const int16_t* l;  // nullptr -> read from global
    __device__ __forceinline__ int at(int idx) const { return l ? (int)l[idx] : (int)g[idx]; }
};

template <int NT = kRowThreads>
__device__ __forceinline__ Labels stage_labels(const LossArgs& a, int16_t* lds) {
    Labels lb;
    lb.g = a.labels;
    lb.l = nullptr;
    const int n = a.B * a.L;
    if (n <= kLabLds) {
        for (int e = threadIdx.x; e < n; e += NT) lds[e] = (int16_t)a.labels[e];
        lb.l = lds;
        __syncthreads();
    }
    return lb;
}

// Locates the jj-th labelled position of sample b (row (b,k) predicts labels[b,k+1]); returns k or -1 and its rank.
// Every wave evaluates this redundantly with ballots (no barrier); the result is workgroup-uniform.
__device__ __forceinline__ int nth_labelled(const Labels& lb, int b, int L, int jj) {
    const int lane = threadIdx.x & 63;
    for (int c0 = 0; c0 < L - 1; c0 += 64) {
        const int k = c0 + lane;
        const bool lab = (k < L - 1) && (lb.at(b * L + k + 1) != -100);
        unsigned long long m = __ballot(lab);
        const int cnt = __popcll(m);
        if (jj < cnt) {
            for (int z = 0; z < jj; ++z) m &= m - 1;  // drop the jj lowest set bits
            return c0 + __ffsll((long long)m) - 1;
        }
        jj -= cnt;
    }
    return -1;
}

// counts of labelled positions: before flat label index `upto` (row-major rank) and in total
template <int NT = kRowThreads>
__device__ __forceinline__ void count_labelled(const Labels& lb, int B, int L, int upto, int& before, int& total, int (*shi)[2]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int cb = 0, ct = 0;
    for (int e = threadIdx.x; e < B * L; e += NT) {
        const int col = e % L;
        const int is = (col != 0 && lb.at(e) != -100) ? 1 : 0;
        ct += is;
        cb += (e < upto) ? is : 0;
    }
    cb = wave_sum(cb);
    ct = wave_sum(ct);
    __syncthreads();
    if (lane == 0) { shi[wv][0] = cb; shi[wv][1] = ct; }
    __syncthreads();
    before = 0;
    total = 0;
    for (int q = 0; q < NT / 64; ++q) { before += shi[q][0]; total += shi[q][1]; }
}


// ---- kernel A: per labelled row: compact rank, logsumexp, label logit, action-slice soft-argmax / argmax ----
template <typename T>
__global__ __launch_bounds__(kRowThreads) void loss_stats_kernel(LossArgs a, int J) {
    // workgroup (j, b) owns the j-th, (j+J)-th, ... labelled position of sample b (usually exactly one row, or none);
    // sample index fastest: consecutive blocks (dealt round-robin to the 8 XCDs) are all active.
    const int j = blockIdx.x / a.B;
    const int b = blockIdx.x - j * a.B;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ int16_t lab16[kLabLds];
    __shared__ float red[16];
    __shared__ int redi[16];
    __shared__ int shi[16][2];
    __shared__ float bmax;
    __shared__ int bidx;
    constexpr int N = Vec<T>::N;
    const int nvec = a.V / N;  // V = 32064 is a multiple of 8
    // one streaming pass: this thread's elements stay in registers (8 f32 or 4 bf16 16-byte vectors = 32 logits)
    constexpr int MAXV = 32 / N;
    float v[MAXV][N];
    const Labels lb = stage_labels(a, lab16);
  for (int jj = j;; jj += J) {
    int rowidx, total;
    const int k = nth_labelled(lb, b, a.L, jj);
    if (k < 0) return;
    const int lab = lb.at(b * a.L + k + 1);
    count_labelled(lb, a.B, a.L, b * a.L + k + 1, rowidx, total, shi);

    const T* z = reinterpret_cast<const T*>(a.logits) + row_offset(a, b, k, rowidx);
    float m = -INFINITY;
    int mi = 0x7fffffff;  // argmax over the vocabulary, lowest index on ties (within a thread columns are visited in increasing order)
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
        for (int e = 0; e < N; ++e)
            if (v[c][e] > m) { m = v[c][e]; mi = q * N + e; }
    }
    for (int q = tid + MAXV * kRowThreads; q < nvec; q += kRowThreads) {  // generic tail for larger vocabularies (re-read from L2)
        float t[N];
        Vec<T>::load(z + (size_t)q * N, t);
#pragma unroll
        for (int e = 0; e < N; ++e)
            if (t[e] > m) { m = t[e]; mi = q * N + e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    __syncthreads();
    if (lane == 0) { red[wv] = m; redi[wv] = mi; }
    __syncthreads();
    if (tid == 0) {
        float mm = red[0];
        int ii = redi[0];
        for (int q = 1; q < kRowThreads / 64; ++q)
            if (red[q] > mm || (red[q] == mm && redi[q] < ii)) { mm = red[q]; ii = redi[q]; }
        bmax = mm;
        bidx = ii;
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

    // the 256 action logits sit in registers of ONE wave: vector index q0 = 31744/N lives at c = q0/1024, tid = q0%1024
    constexpr int q0 = kA0 / N, cs = q0 / kRowThreads, t0 = q0 % kRowThreads, nthr = kNA / N;
    static_assert(t0 % 64 == 0 && nthr <= 64 && cs < MAXV, "action slice must map onto one wave");
    float alse = 0.0f, E = 0.0f;
    int pred = 0;
    if (wv == t0 / 64) {
        const bool own = lane < nthr;
        float x[N];
#pragma unroll
        for (int e = 0; e < N; ++e) x[e] = own ? v[cs][e] : -INFINITY;
        float am = x[0];
        int ai = 0;
#pragma unroll
        for (int e = 1; e < N; ++e) if (x[e] > x[ai]) ai = e;
        float bestv = x[ai];
        int besti = lane * N + ai;
#pragma unroll
        for (int e = 1; e < N; ++e) am = fmaxf(am, x[e]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {  // argmax with lowest-index tie break (torch.argmax)
            const float ov = __shfl_xor(bestv, o, 64);
            const int oi = __shfl_xor(besti, o, 64);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        am = wave_max(am);
        float es = 0.0f, ew = 0.0f;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const float ex = expf(x[e] - am);
            es += ex;
            ew += ex * (float)(lane * N + e + 1);
        }
        es = wave_sum(es);
        ew = wave_sum(ew);
        alse = am + logf(es);
        E = ew / es;
        pred = kA0 + besti;
    }
    __syncthreads();
    if (tid == t0) {  // first lane of the slice wave
        float tot = 0.0f;
        for (int q = 0; q < kRowThreads / 64; ++q) tot += red[q];
        RowStat r;
        r.lse = M + logf(tot);
        r.zlab = Vec<T>::get(z + lab);
        r.alse = alse;
        r.E = E;
        r.pred = pred;
        r.pos = b * (a.L - 1) + k;
        r.lab = lab;
        r.ord = jj;
        r.predf = bidx;
        a.st[rowidx] = r;
    }
  }
}


// ---- kernel B: every labelled row folds the compact per-row statistics into the global scalars it needs (fixed order ->
//      deterministic and identical in all workgroups), then writes its gradient row; rank-0's workgroup (or workgroup 0
//      when nothing is labelled) publishes the scalars and the predicted tokens. ----
constexpr int kGradT = 512;  // 64 logits per thread stay in registers (256-VGPR budget), fp64 reductions do not spill

template <typename T>
__global__ __launch_bounds__(kGradT) void loss_grad_kernel(LossArgs a, int J) {
    const int j = blockIdx.x / a.B, b = blockIdx.x - j * a.B;
    const int tid = threadIdx.x;
    __shared__ int16_t lab16[kLabLds];
    __shared__ int shi[kGradT / 64][2];
    __shared__ double sh[kGradT / 64][7];
    const Labels lb = stage_labels<kGradT>(a, lab16);
  for (int jj = j;; jj += J) {
    const int k = nth_labelled(lb, b, a.L, jj);
    if (k < 0 && !(blockIdx.x == 0 && jj == j)) return;  // workgroup 0 always runs once: it publishes zeros when nothing is labelled
    int rowidx, R;
    RowStat me;
    me.lse = me.alse = me.E = 0.0f; me.ord = 0; me.lab = -100;
    count_labelled<kGradT>(lb, a.B, a.L, k < 0 ? 0 : b * a.L + k + 1, rowidx, R, shi);
    const int lab = k < 0 ? -100 : lb.at(b * a.L + k + 1);

    // this row's logits: issue the loads now, use them after the reductions (addresses do not depend on the statistics)
    constexpr int N = Vec<T>::N;
    constexpr int MAXV = 64 / N;
    const int nvec = a.V / N;
    const size_t off = k < 0 ? 0 : row_offset(a, b, k, rowidx);
    const T* z = reinterpret_cast<const T*>(a.logits) + off;
    const bool want_row = (k >= 0) && a.glogits && (a.mode != VAA_LOSS_UADA_DDP && a.mode != VAA_LOSS_UPA);  // CE term needs every logit

    double acc[7] = {0, 0, 0, 0, 0, 0, 0};  // ce, mse, uad, nrow, nact, upa sum(cos+1), upa sum ||e'-l'||
    for (int r = tid; r < R; r += kGradT) {
        const RowStat s = a.st[r];
        acc[3] += 1.0;
        acc[0] += (double)s.lse - (double)s.zlab;
        if (s.lab > 2) {
            acc[4] += 1.0;
            const double rr = (double)s.E / 256.0, t = (s.lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10: 1/256 -> 0)
            acc[1] += (rr - t) * (rr - t);
            const double ag = bin_center(s.lab), ap = bin_center(s.pred);  // cal_UAD, UADA.py:408-418
            acc[2] += fabs(ap - ag) / (ag > 0 ? fabs(ag + 1.0) : fabs(ag - 1.0));
        }
        if (a.mode == VAA_LOSS_UPA && s.ord == 0 && r + 2 < R) {  // first three labelled rows of a sample are consecutive ranks
            Upa3 u;
            u.set(0, s); u.set(1, a.st[r + 1]); u.set(2, a.st[r + 2]);
            double c1, nd;
            u.terms(c1, nd);
            acc[5] += c1;
            acc[6] += nd;
        }
    }
    block_sums<7, kGradT>(acc, sh);
    const double nrow = acc[3], nact = acc[4];
    const double CE = nrow > 0 ? acc[0] / nrow : 0.0;
    const double MSE = nact > 0 ? (double)a.w * a.w * acc[1] / nact : 0.0;
    const double UAD = nact > 0 ? acc[2] / nact : 0.0;

    double total = 0.0, aux0 = 0.0, aux1 = 0.0;
    float kce = 0.0f, kE = 0.0f;
    if (k >= 0) me = a.st[rowidx];
    if (a.mode == VAA_LOSS_UPA) {
        aux0 = acc[5] / a.B;
        aux1 = 1.0 / (acc[6] / a.B + 1e-3);  // UPA.py:384
        total = (double)a.alpha * aux0 + (double)a.beta * aux1;
        if (k >= 0 && me.ord < 3 && rowidx - me.ord + 2 < R) {
            Upa3 u;
            u.set(0, a.st[rowidx - me.ord]); u.set(1, a.st[rowidx - me.ord + 1]); u.set(2, a.st[rowidx - me.ord + 2]);
            kE = (float)(u.dE(me.ord, (double)a.alpha, (double)a.beta, aux1, a.B) / 255.0);
        }
    } else {
        double dce = 0.0;
        if (a.mode == VAA_LOSS_UADA) { total = MSE + 1.0 / CE; dce = -1.0 / (CE * CE); }  // UADA.py:147
        else if (a.mode == VAA_LOSS_UADA_DDP) { total = MSE; }                            // UADA_ddp.py:203-206
        else { total = (double)a.scale * CE; dce = (double)a.scale; }                     // TMA.py:148
        kce = nrow > 0 ? (float)(dce / nrow) : 0.0f;
        if (a.mode != VAA_LOSS_CE && lab > 2) {
            const double rr = (double)me.E / 256.0, t = (lab > 31872) ? 0.0 : 1.0;
            kE = (float)((double)a.w * a.w * 2.0 * (rr - t) / nact / 256.0);
        }
    }

    // publication: scalars by the rank-0 row (or by workgroup 0 when nothing is labelled); predicted tokens likewise
    const bool first = (k >= 0) ? (rowidx == 0) : (R == 0);
    if (first && tid == 0) {
        a.scalars[0] = (float)total; a.scalars[1] = (float)CE; a.scalars[2] = (float)MSE; a.scalars[3] = (float)aux0;
        a.scalars[4] = (float)aux1; a.scalars[5] = (float)nrow; a.scalars[6] = (float)nact; a.scalars[7] = (float)UAD;
    }
    if (first && (a.pred_tokens || a.pred_full)) {
        const int P = a.B * (a.L - 1);
        for (int q = tid; q < P; q += kGradT) {
            if (a.pred_tokens) a.pred_tokens[q] = -1;
            if (a.pred_full) a.pred_full[q] = -1;
        }
        __syncthreads();
        for (int r = tid; r < R; r += kGradT) {
            const RowStat s = a.st[r];
            if (a.pred_tokens && s.lab > 2) a.pred_tokens[s.pos] = s.pred;
            if (a.pred_full) a.pred_full[s.pos] = s.predf;
        }
    }
    if (k < 0) return;
    if (!a.glogits) continue;

    T* g = reinterpret_cast<T*>(a.glogits) + off;
    float x[MAXV][N];
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {  // all of this thread's loads first, then the math
        const int q = tid + c * kGradT;
        const int v0 = q * N;
        const bool in_slice = (v0 >= kA0 && v0 < kA0 + kNA);
        if (q < nvec && (want_row || (in_slice && kE != 0.0f))) Vec<T>::load(z + (size_t)v0, x[c]);
    }
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = tid + c * kGradT;
        if (q >= nvec) continue;
        const int v0 = q * N;
        const bool in_slice = (v0 >= kA0 && v0 < kA0 + kNA);
        float o[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float gv = 0.0f;
            if (kce != 0.0f) gv = kce * (expf(x[c][e] - me.lse) - ((v0 + e) == lab ? 1.0f : 0.0f));
            if (in_slice && kE != 0.0f) {
                const float pa = expf(x[c][e] - me.alse);
                gv += kE * pa * ((float)(v0 + e - kA0 + 1) - me.E);
            }
            o[e] = gv;
        }
        Vec<T>::store(g + (size_t)v0, o);
    }
    for (int q = tid + MAXV * kGradT; q < nvec; q += kGradT) {  // generic tail for larger vocabularies
        float t[N], o[N];
        const int v0 = q * N;
        if (kce != 0.0f) Vec<T>::load(z + (size_t)v0, t);
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = kce != 0.0f ? kce * (expf(t[e] - me.lse) - ((v0 + e) == lab ? 1.0f : 0.0f)) : 0.0f;
        Vec<T>::store(g + (size_t)v0, o);
    }
  }
}

// =====================================================================================================================
// ROWS fast path (what forward_rows feeds): logits [R,V] of the labelled rows only, with a ROW MAP built once per outer
// iteration (labels do not change during the innerLoop steps: UADA.py:130-133), so a step never touches the label matrix.
//   * rows_stats_kernel : grid = R x SPLIT workgroups of 512 threads; workgroup (r, h) streams 1/SPLIT of row r (so R' = 128
//     rows fill all 256 CUs), reduces {max, argmax, sum-exp, label logit} of its part; the part that holds the 256 action
//     logits also produces the soft-argmax statistics and — in UADA_DDP mode, whose gradient needs nothing from other rows —
//     writes the row's gradient slice at once.
//   * rows_finish_kernel: combines the parts, folds the R compact statistics into the scalars (fixed order -> deterministic),
//     publishes scalars + predicted tokens (slice argmax for UAD, UADA.py:395, and full-vocabulary argmax for the metrics of
//     UADA.py:165-167 / TMA.py:148-149) and writes the gradients that depend on global scalars (1/CE, CE, UPA).
//   Gradient storage: VAA_GRAD_FULL [R,V] or VAA_GRAD_SLICE [R,256] (UADA_DDP / UPA: the gradient is zero outside the action
//   columns, so the LM-head backward contracts over 256 columns instead of 32,064).
// =====================================================================================================================
// RowMap / PartStat / SliceStat: vaa_rows.h (shared with the fused LM head of vaa_head.hip)

// threads per workgroup of the row kernels: 256 while a row fits 4 parts x 256 threads x 32 logits (V <= 32,768: fewer waves per barrier,
// 512 workgroups at R' = 128 — 15.9 -> 14.6 us), else 512
constexpr int kRowsTMax = 512;
static int rows_threads(int V) { return V <= 4 * 256 * 32 ? 256 : 512; }

__global__ __launch_bounds__(1024) void loss_rowmap_kernel(const int64_t* __restrict__ labels, int B, int L, int* __restrict__ out) {
    // out: int hdr[4] = {R, n_action_rows, 0, 0}, then RowMap[R] in (b,k) row-major order of labels[b,k+1] != -100
    __shared__ int wsum[16][2];
    __shared__ int carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    RowMap* rm = reinterpret_cast<RowMap*>(out + 4);
    if (tid == 0) { carry[0] = 0; carry[1] = 0; }
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        int cnt = 0, act = 0;
        if (b < B)
            for (int e = 1; e < L; ++e) {
                const long v = labels[(size_t)b * L + e];
                cnt += (v != -100) ? 1 : 0;
                act += (v > 2) ? 1 : 0;
            }
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        const int wact = wave_sum(act);
        if (lane == 63) wsum[wv][0] = inc;
        if (lane == 0) wsum[wv][1] = wact;
        __syncthreads();
        int base = carry[0], tot = 0, tact = 0;
        for (int q = 0; q < 16; ++q) { if (q < wv) base += wsum[q][0]; tot += wsum[q][0]; tact += wsum[q][1]; }
        if (b < B) {
            int r = base + inc - cnt, ord = 0;
            for (int e = 1; e < L; ++e) {
                const long v = labels[(size_t)b * L + e];
                if (v != -100) { RowMap m; m.b = b; m.k = e - 1; m.lab = (int)v; m.ord = ord++; rm[r++] = m; }
            }
        }
        __syncthreads();
        if (tid == 0) { carry[0] += tot; carry[1] += tact; }
        __syncthreads();
    }
    if (tid == 0) { out[0] = carry[0]; out[1] = carry[1]; out[2] = 0; out[3] = 0; }
}


template <typename T>
__device__ __forceinline__ void store_slice_or_row(const RowsArgs& a, int r, int col0, const float* o) {  // Vec<T>::N values at column col0
    T* g = reinterpret_cast<T*>(a.grad);
    if (a.grad_slice) Vec<T>::store(g + (size_t)r * kNA + (col0 - kA0), o);
    else Vec<T>::store(g + (size_t)r * a.V + col0, o);
}


struct SliceStat;
template <typename T, int kRowsT>
__device__ __forceinline__ void rows_full_gradient(const RowsArgs& a, float kce, double nact, int r, int v_lo, int v_hi,
                                                   const float (&v)[32 / Vec<T>::N][Vec<T>::N], float lse, float alse, float E);


// ONEPASS (full-row gradients of UADA / CE, whose scale needs the folded scalars): the statistics pass keeps its 32 logits per thread in
// registers across a grid-wide barrier, folds the statistics and writes the gradient from them — the logits are read ONCE (the two-launch
// form reads every row again in its finishing launch). bar: two zeroed words owned by this launch's stream, left zero.
template <typename T, int kRowsT, bool ONEPASS>
__global__ __launch_bounds__(kRowsT) void rows_stats_kernel(RowsArgs a, unsigned* bar, unsigned gen, unsigned* err_word, int max_polls) {
    constexpr int N = Vec<T>::N;
    constexpr int MAXV = 32 / N;  // 32 logits per thread in registers: covers V/split <= 16384
    const int r = blockIdx.x / a.split, h = blockIdx.x - r * a.split;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nvec = a.V / N, pvec = (nvec + a.split - 1) / a.split;  // vectors per part
    const int v_lo = h * pvec, v_hi = min(nvec, v_lo + pvec);
    const T* z = reinterpret_cast<const T*>(a.logits) + (size_t)r * a.V;
    __shared__ float redf[kRowsT / 64];
    __shared__ int redi[kRowsT / 64];
    __shared__ float bmax;
    __shared__ int bidx;
    float v[MAXV][N];
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = v_lo + tid + c * kRowsT;
        if (q < v_hi) {
            Vec<T>::load(z + (size_t)q * N, v[c]);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) v[c][e] = -INFINITY;
        }
    }
    // the action slice, for the part that holds it: one extra 16-byte load per lane of wave 0 (L1/L2 hit)
    const bool has_slice = (kA0 / N) >= v_lo && (kA0 / N) < v_hi;
    constexpr int nthr = kNA / N;
    float x[N];
    if (has_slice && wv == 0 && lane < nthr) Vec<T>::load(z + kA0 + lane * N, x);
    // a caller row count beyond the map's own count (vaa.h: scalars[0] = NaN then) must not index unbuilt map entries
    RowMap me = {0, 0, -1, 0};
    if (r < a.rowmap[0]) me = reinterpret_cast<const RowMap*>(a.rowmap + 4)[r];
    const int nact = a.rowmap[1];

    // the label's logit (one thread): requested here, behind the row loads, not after the reductions
    float zlab_early = -INFINITY;
    if (tid == 0) {
        const int lv = me.lab / N;
        if (me.lab >= 0 && me.lab < a.V && lv >= v_lo && lv < v_hi) zlab_early = Vec<T>::get(z + me.lab);
    }
    // ---- max + argmax (lowest index on ties, torch.argmax) ----
    float m = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = v_lo + tid + c * kRowsT;
#pragma unroll
        for (int e = 0; e < N; ++e)
            if (v[c][e] > m) { m = v[c][e]; mi = q * N + e; }  // within a thread columns are visited in increasing order
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { redf[wv] = m; redi[wv] = mi; }
    __syncthreads();
    if (tid == 0) {
        float mm = redf[0];
        int ii = redi[0];
        for (int q = 1; q < kRowsT / 64; ++q)
            if (redf[q] > mm || (redf[q] == mm && redi[q] < ii)) { mm = redf[q]; ii = redi[q]; }
        bmax = mm;
        bidx = ii;
    }
    __syncthreads();
    const float M = bmax;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c)
#pragma unroll
        for (int e = 0; e < N; ++e) s += expf(v[c][e] - M);  // exp(-inf) = 0 for the padding
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) redf[wv] = s;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.0f;
        for (int q = 0; q < kRowsT / 64; ++q) tot += redf[q];
        PartStat ps;
        ps.m = M;
        ps.s = tot;
        ps.amax = bidx;
        ps.zlab = zlab_early;
        stat_store<ONEPASS>(&a.part[(size_t)r * a.split + h], ps);
    }
    if (!ONEPASS && !(has_slice && wv == 0)) return;
    if (has_slice && wv == 0) {
    // ---- action slice: soft-argmax statistics by one wave (UADA.py:384-389, UPA.py:370-374) ----
    const bool own = lane < nthr;
    if (!own) {
#pragma unroll
        for (int e = 0; e < N; ++e) x[e] = -INFINITY;
    }
    int ai = 0;
#pragma unroll
    for (int e = 1; e < N; ++e) if (x[e] > x[ai]) ai = e;
    float bestv = x[ai], am = x[ai];
    int besti = lane * N + ai;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
    }
    am = wave_max(am);
    float ex[N], es = 0.0f, ew = 0.0f;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        ex[e] = expf(x[e] - am);
        es += ex[e];
        ew += ex[e] * (float)(lane * N + e + 1);
    }
    es = wave_sum(es);
    ew = wave_sum(ew);
    const float E = ew / es;
    if (lane == 0) {
        SliceStat ss;
        ss.alse = am + logf(es);
        ss.E = E;
        ss.pred = kA0 + besti;
        ss.pad = 0;
        stat_store<ONEPASS>(&a.slice[r], ss);
    }
    if (a.mode == VAA_LOSS_UADA_DDP && a.grad && own) {  // gradient of w^2*mean((E/256 - t)^2): needs this row and the row COUNT only
        float o[N];
        float kE = 0.0f;
        if (me.lab > 2 && nact > 0) {
            const double rr = (double)E / 256.0, t = (me.lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10)
            kE = (float)((double)a.w * a.w * 2.0 * (rr - t) / nact / 256.0);
        }
        const float alse = am + logf(es);
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = kE * expf(x[e] - alse) * ((float)(lane * N + e + 1) - E);
        store_slice_or_row<T>(a, r, kA0 + lane * N, o);
    }
    }
    if (!ONEPASS) return;
    // ---- grid-wide hand-over without cache-wide fences: every cross-workgroup word travels through agent-scope atomics (they bypass the
    // per-XCD L2). Arrival is counted in two levels, each counter on its own 128-byte line — eight counters by blockIdx & 7, then one — so no
    // line takes more than nwg / 8 read-modify-writes. The LAST workgroup folds the statistics once, leaves every row's log-sum-exp beside
    // its slice statistics and publishes {kce, nact, Rn} as two self-validating 64-bit words {generation, payload} on a line of their own;
    // everybody else polls those two words (two lanes, one instruction per round), then fetches ONE 16-byte record of its own row.
    // What was measured on the way, R' = 128, bf16 (two launches: 20.3 us): release/acquire fences per workgroup 42 us; ticket, flag, leave
    // count and results on one line 33 us; separate lines + a flag + a fetch of seven words and the row's record 19.6 us; a 64-byte record
    // per row polled by 16 lanes 22.3 us; row cells as the first arrival level + the row's statistics fetched while waiting 20.8 us (every
    // extra polling lane costs more than the round trip it saves); this form 20.0 us (fp32, R' = 256: 25.8 against 30.2). ----
    __shared__ int bar_ok, am_last;
    __shared__ double shf[kRowsT / 64][7];
    __shared__ unsigned hand[16];  // 0: kce, 1: nact | Rn << 16; 8..11: the row's {alse, E, pred, log-sum-exp}
    // this thread's statistics stores (agent-scope atomic stores above) must have COMPLETED before the arrival below is counted: a
    // workgroup-scope release fence only waits for LDS / scalar traffic (lgkmcnt) in non-tgsplit mode, so the vector-memory counter is
    // drained explicitly (round 3 shipped without it: the folding workgroup could read a stale PartStat / SliceStat — ADVICE r3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    unsigned long long* pub = reinterpret_cast<unsigned long long*>(bar + 32 * 9);
    if (tid == 0) {
        const unsigned nwg = gridDim.x, sub = blockIdx.x & 7u;
        const unsigned nres = (nwg + 7u - sub) >> 3, ntop = nwg < 8u ? nwg : 8u;
        unsigned* subc = bar + 32 * (1 + sub);
        bool last = false;
        if (atomicAdd(subc, 1u) == nres - 1u) {  // everybody of this residue has arrived: re-arm the counter, report upwards
            __hip_atomic_store(subc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (atomicAdd(bar, 1u) == ntop - 1u) {
                __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = true;
            }
        }
        am_last = last ? 1 : 0;
        bar_ok = 1;
    }
    __syncthreads();
    // coherent loads; scalars[8] + the prediction maps like rows_finish_kernel's block 0
    if (am_last) (void)rows_fold<kRowsT, true>(a, true, shf, Handover{pub, gen});
    if (wv == 0) {
        unsigned long long w = 0ull;
        int it = 0;
        for (;;) {  // (the folding workgroup finds its own words at once)
            if (lane < 2) w = __hip_atomic_load(&pub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane >= 2 || (unsigned)(w >> 32) == gen;
            if (__all(ok)) break;                         // wave-uniform
            if (++it > max_polls) { it = -1; break; }    // gave up
            __builtin_amdgcn_s_sleep(4);
        }
        if (lane < 2) hand[lane] = (unsigned)w;
        // a launch that never became fully resident gives up instead of hanging (the host admits resident grids only): NaN gradient AND the
        // process-wide failure word (pinned host memory): the next library call returns VAA_E_LAUNCH (vaa_async_error)
        if (lane == 0) {
            bar_ok = it >= 0;
            if (it < 0 && err_word) __hip_atomic_store(err_word, VAA_ASYNC_K3_HANDOVER_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // a plain store: no PCIe atomics needed
        }
        if (lane >= 8 && lane < 12) hand[lane] = __hip_atomic_load(reinterpret_cast<const unsigned*>(&a.slice[r]) + (lane - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    float kce = __uint_as_float(hand[0]);
    int nact_all = (int)(hand[1] & 0xffffu), Rn = (int)(hand[1] >> 16);
    if (!bar_ok) {  // the words never came: every row of the launch gets a NaN gradient (never a stale one, never nothing)
        kce = __uint_as_float(0x7fc00000u);
        Rn = a.R;
        nact_all = 1;
    }
    if (a.grad && r < Rn) rows_full_gradient<T, kRowsT>(a, kce, (double)nact_all, r, v_lo, v_hi, v, __uint_as_float(hand[11]), __uint_as_float(hand[8]), __uint_as_float(hand[9]));
}

// d total / d logits of row r, part [v_lo, v_hi), from the logits a thread holds (v[c] = vector v_lo + tid + c * kRowsT): full-row modes
// (UADA: 1/CE^2 term + the action-slice term, CE).
template <typename T, int kRowsT>
__device__ __forceinline__ void rows_full_gradient(const RowsArgs& a, float kce, double nact, int r, int v_lo, int v_hi,
                                                   const float (&v)[32 / Vec<T>::N][Vec<T>::N], float lse, float alse, float E) {
    constexpr int N = Vec<T>::N;
    constexpr int MAXV = 32 / N;
    const int tid = threadIdx.x;
    const RowMap me = reinterpret_cast<const RowMap*>(a.rowmap + 4)[r];
    struct { float alse, E; } ms = {alse, E};
    float kE = 0.0f;
    if (a.mode != VAA_LOSS_CE && me.lab > 2) {
        const double q = (double)ms.E / 256.0, t = (me.lab > 31872) ? 0.0 : 1.0;
        kE = (float)((double)a.w * a.w * 2.0 * (q - t) / nact / 256.0);
    }
    T* g = reinterpret_cast<T*>(a.grad) + (size_t)r * a.V;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = v_lo + tid + c * kRowsT;
        if (q >= v_hi) continue;
        const int v0 = q * N;
        const bool in_slice = (v0 >= kA0 && v0 < kA0 + kNA);
        float o[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float gv = 0.0f;
            if (kce != 0.0f) gv = kce * (expf(v[c][e] - lse) - ((v0 + e) == me.lab ? 1.0f : 0.0f));
            if (in_slice && kE != 0.0f) gv += kE * expf(v[c][e] - ms.alse) * ((float)(v0 + e - kA0 + 1) - ms.E);
            o[e] = gv;
        }
        Vec<T>::store(g + (size_t)v0, o);
    }
}


// grid = R x split (full-row gradients) or R (slice / no gradient); every workgroup folds the statistics in the same fixed order.
// fold_wg >= 0 (VAA_LOSS_CE with a full-row gradient: d total / d z = scale / nrow (softmax - onehot) needs the row's own parts and the row COUNT
// of the map, nothing of other rows): workgroup fold_wg — one past the gradient workgroups — folds and publishes the scalars, every other
// workgroup goes straight to its gradient (same parts, same formulas: the bits of the all-fold form) instead of waiting for a fold it does not use.
template <typename T, int kRowsT>
__global__ __launch_bounds__(kRowsT) void rows_finish_kernel(RowsArgs a, int gsplit, int fold_wg) {
    constexpr int N = Vec<T>::N;
    const int r = blockIdx.x / gsplit, h = blockIdx.x - r * gsplit;
    const int tid = threadIdx.x;
    __shared__ double sh[kRowsT / 64][7];
    const RowMap* rm = reinterpret_cast<const RowMap*>(a.rowmap + 4);
    // this workgroup's part of its row: issue the loads before the fold (addresses do not depend on it)
    const bool full_grad = a.grad && !a.grad_slice && (a.mode == VAA_LOSS_UADA || a.mode == VAA_LOSS_CE);
    const bool zero_fill = a.grad && !a.grad_slice && !full_grad;  // FULL storage asked for a slice-only mode: zeros outside the slice
    constexpr int MAXV = 32 / N;
    const int nvec = a.V / N, pvec = (nvec + gsplit - 1) / gsplit;
    const int v_lo = h * pvec, v_hi = min(nvec, v_lo + pvec);
    const T* z = reinterpret_cast<const T*>(a.logits) + (ptrdiff_t)r * a.ldz - a.zcol0;
    float v[MAXV][N];
    if (full_grad && r < a.R) {
#pragma unroll
        for (int c = 0; c < MAXV; ++c) {
            const int q = v_lo + tid + c * kRowsT;
            if (q < v_hi) Vec<T>::load(z + (size_t)q * N, v[c]);
        }
    }
    auto row_lse = [&](int rr, float& zlab, int& amax) {  // combine the parts of row rr
        float M = -INFINITY;
        for (int q = 0; q < a.split; ++q) M = fmaxf(M, a.part[(size_t)rr * a.split + q].m);
        float tot = 0.0f, best = -INFINITY;
        zlab = -INFINITY;
        amax = 0x7fffffff;
        for (int q = 0; q < a.split; ++q) {
            const PartStat p = a.part[(size_t)rr * a.split + q];
            tot += p.s * expf(p.m - M);
            zlab = fmaxf(zlab, p.zlab);
            if (p.m > best || (p.m == best && p.amax < amax)) { best = p.m; amax = p.amax; }
        }
        return M + logf(tot);
    };
    auto upa_of = [&](int r0, Upa3& u) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            RowStat t;
            t.E = a.slice[r0 + q].E;
            t.lab = rm[r0 + q].lab;
            u.set(q, t);
        }
    };
    FoldOut f;
    if (fold_wg < 0) {
        f = rows_fold<kRowsT>(a, blockIdx.x == 0, sh);
    } else if ((int)blockIdx.x == fold_wg) {
        (void)rows_fold<kRowsT>(a, true, sh);
        return;
    } else {  // what rows_fold returns for VAA_LOSS_CE, as far as the gradient reads it
        f.Rn = min(a.R, a.rowmap[0]);
        f.nrow = (double)f.Rn;
        f.nact = 0.0; f.aux1 = 0.0;
        f.dce = (double)a.scale;
    }
    const int Rn = f.Rn;
    const double nrow = f.nrow, nact = f.nact, dce = f.dce, aux1 = f.aux1;
    if (!a.grad || r >= Rn) return;
    const RowMap me = rm[r];
    if (a.mode == VAA_LOSS_UADA_DDP && !zero_fill) return;  // slice already written by the statistics kernel
    // ---- gradient of this row (part h) ----
    const SliceStat ms = a.slice[r];
    float kce = nrow > 0 ? (float)(dce / nrow) : 0.0f, kE = 0.0f;
    if (a.mode == VAA_LOSS_UPA) {
        kce = 0.0f;
        if (me.ord < 3 && r - me.ord >= 0 && r - me.ord + 2 < Rn) {
            Upa3 u;
            upa_of(r - me.ord, u);
            kE = (float)(u.dE(me.ord, (double)a.alpha, (double)a.beta, aux1, a.B) / 255.0);
        }
    } else if (a.mode != VAA_LOSS_CE && me.lab > 2) {
        const double q = (double)ms.E / 256.0, t = (me.lab > 31872) ? 0.0 : 1.0;
        kE = (float)((double)a.w * a.w * 2.0 * (q - t) / nact / 256.0);
    }
    if (!full_grad) {  // slice-only modes (UPA; UADA_DDP only when FULL storage was asked for)
        constexpr int nthr = kNA / N;
        if (zero_fill) {
            float o[N];
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = 0.0f;
            for (int q = v_lo + tid; q < v_hi; q += kRowsT)
                if (q * N < kA0 || q * N >= kA0 + kNA) Vec<T>::store(reinterpret_cast<T*>(a.grad) + (size_t)r * a.V + (size_t)q * N, o);
        }
        const bool owns_slice = (kA0 / N) >= v_lo && (kA0 / N) < v_hi;
        if (owns_slice && tid < nthr) {
            float x[N], o[N];
            Vec<T>::load(z + kA0 + tid * N, x);
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = kE * expf(x[e] - ms.alse) * ((float)(tid * N + e + 1) - ms.E);
            store_slice_or_row<T>(a, r, kA0 + tid * N, o);
        }
        return;
    }
    float zl;
    int am;
    const float lse = row_lse(r, zl, am);
    T* g = reinterpret_cast<T*>(a.grad) + (size_t)r * a.V;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int q = v_lo + tid + c * kRowsT;
        if (q >= v_hi) continue;
        const int v0 = q * N;
        const bool in_slice = (v0 >= kA0 && v0 < kA0 + kNA);
        float o[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float gv = 0.0f;
            if (kce != 0.0f) gv = kce * (expf(v[c][e] - lse) - ((v0 + e) == me.lab ? 1.0f : 0.0f));
            if (in_slice && kE != 0.0f) gv += kE * expf(v[c][e] - ms.alse) * ((float)(v0 + e - kA0 + 1) - ms.E);
            o[e] = gv;
        }
        Vec<T>::store(g + (size_t)v0, o);
    }
}

// Step epilogue (vaa_step_epilogue): ONE launch between the backward and the gradient exchange.
//   workgroups 0 .. nred-1 : msg[e] = sum over K2's partial tiles, the fixed order of patch_grad_reduce_kernel (bitwise the same result)
//   workgroup  nred        : K3's statistics folded into scalars[8] + the prediction maps (what rows_finish_kernel does in the slice modes
//                            whose gradient does not wait for it), then the tail of the sync message msg[n..n+4) = {CE, w^2*MSE, UAD, total}
struct EpiArgs {
    const float* partials;
    float* msg;
    const float* scalars_in;  // fold == 0: final scalars of an earlier vaa_loss_rows_fwd_bwd, only copied into the message
    int n, nparts, nred, fold;
    int fuse_update;          // single-GPU step: K4's per-element update applied right where the gradient element is produced
    UpdArgs upd;
    double* stat_part;        // fuse_update: [nred][2] = {sum |g|, sum g} over the block's 64 elements (the caller adds them for the log)
};

__global__ __launch_bounds__(256) void step_epilogue_kernel(EpiArgs e, RowsArgs a) {
    __shared__ double sl[16][16][4];
    if ((int)blockIdx.x < e.nred) {
        int oe = 0;
        float g = 0.0f;
        const bool own = partial_reduce_block(e.partials, e.msg, e.n, e.nparts, blockIdx.x, sl, oe, g);
        if (!e.fuse_update) return;
        // K4 on the element this thread just produced (grad_scale = 1, no L1 clip: vaa_step_epilogue_update checks): same arithmetic, same bits
        if (own) {
            float m = e.upd.mode == VAA_OPT_ADAMW_HF ? e.upd.m[oe] : 0.0f, v = e.upd.mode == VAA_OPT_ADAMW_HF ? e.upd.v[oe] : 0.0f;
            const float p = update_one(e.upd, g, e.upd.patch[oe], m, v);
            if (e.upd.mode == VAA_OPT_ADAMW_HF) { e.upd.m[oe] = m; e.upd.v[oe] = v; }
            e.upd.patch[oe] = p;
        }
        if (e.stat_part) {  // the owners are threads 0..63 (slices 0..3 x 16 quads) = wave 0
            double sa = own ? fabs((double)g) : 0.0, ss = own ? (double)g : 0.0;
            if (threadIdx.x < 64) {
                sa = wave_sum(sa);
                ss = wave_sum(ss);
                if (threadIdx.x == 0) { e.stat_part[2 * blockIdx.x] = sa; e.stat_part[2 * blockIdx.x + 1] = ss; }
            }
        }
        return;
    }
    float* tail = e.msg + e.n;
    if (e.fold) {
        double (*sh)[7] = reinterpret_cast<double (*)[7]>(&sl[0][0][0]);
        const FoldOut f = rows_fold<256>(a, true, sh);
        if (threadIdx.x == 0) {
            const bool ok = a.rowmap[0] == a.R;
            tail[0] = (float)f.CE; tail[1] = (float)f.MSE; tail[2] = (float)f.UAD;
            tail[3] = ok ? (float)f.total : __uint_as_float(0x7fc00000u);
        }
    } else if (threadIdx.x == 0) {
        tail[0] = e.scalars_in[1]; tail[1] = e.scalars_in[2]; tail[2] = e.scalars_in[7]; tail[3] = e.scalars_in[0];
    }
}

int rows_split(int R, int V) {  // parts per row so that >= 256 workgroups are resident; a part must fit threads x 32 logits
    const int nt = rows_threads(V);
    int s = 1;
    while (s < 4 && R * s < 256) s <<= 1;
    while ((V + s - 1) / s > nt * 32) s <<= 1;
    return s;
}

}  // namespace vaa

extern "C" size_t vaa_loss_rowmap_bytes(int B, int L);
extern "C" size_t vaa_loss_rows_ws_bytes(int R);
extern "C" int vaa_loss_rowmap_build(const int64_t* labels, int B, int L, void* rowmap, size_t rowmap_bytes, void* stream);
extern "C" int vaa_loss_rows_fwd_bwd(const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode,
                                     const float* params, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* grad,
                                     int grad_kind, void* ws, size_t ws_bytes, void* stream);

static size_t align256(size_t n) { return (n + 255) / 256 * 256; }

extern "C" size_t vaa_loss_ws_bytes(int B, int L) {
    if (B <= 0 || L <= 1) return 0;
    const size_t legacy = (size_t)B * (size_t)(L - 1) * sizeof(vaa::RowStat);
    const size_t rows = align256(vaa_loss_rowmap_bytes(B, L)) + vaa_loss_rows_ws_bytes(B * (L - 1));  // ROWS layout: row map + row statistics
    return legacy > rows ? legacy : rows;
}

extern "C" int vaa_loss_fwd_bwd_ex(const void* logits, int dtype, int layout, const int64_t* labels, int B, int S, int L, int V,
                                   int mode, const float* params, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens,
                                   void* glogits, void* ws, size_t ws_bytes, void* stream) {
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
    if (layout == VAA_LAYOUT_ROWS && S > 0) {
        // row count given: build the row map in the workspace and take the row-split path (callers that keep the labels for many
        // steps build the map once with vaa_loss_rowmap_build and call vaa_loss_rows_fwd_bwd directly)
        const size_t map_bytes = align256(vaa_loss_rowmap_bytes(B, L));
        int rc0 = vaa_loss_rowmap_build(labels, B, L, ws, map_bytes, stream);
        if (rc0 != VAA_OK) return rc0;
        return vaa_loss_rows_fwd_bwd(logits, dtype, ws, S, B, L, V, mode, params, scalars, pred_tokens, pred_full_tokens, glogits, VAA_GRAD_FULL,
                                     (char*)ws + map_bytes, ws_bytes - map_bytes, stream);
    }
    LossArgs a;
    a.logits = logits; a.labels = labels; a.st = (RowStat*)ws; a.glogits = glogits; a.scalars = scalars; a.pred_tokens = pred_tokens;
    a.pred_full = pred_full_tokens;
    a.B = B; a.S = S; a.L = L; a.V = V; a.mode = mode; a.layout = layout;
    a.w = params[0]; a.alpha = params[1]; a.beta = params[2]; a.scale = params[3];
    const int J = (L - 1) < 8 ? (L - 1) : 8;  // workgroups per sample; the attacks label at most 8 positions per sample
    const unsigned G = (unsigned)B * (unsigned)J;
    if (dtype == VAA_DTYPE_F32) VAA_LAUNCH(loss_stats_kernel<float>, dim3(G), dim3(kRowThreads), 0, st, a, J);
    else VAA_LAUNCH(loss_stats_kernel<uint16_t>, dim3(G), dim3(kRowThreads), 0, st, a, J);
    int rc = check_launch("vaa_loss_fwd_bwd(stats)");
    if (rc != VAA_OK) return rc;
    if (dtype == VAA_DTYPE_F32) VAA_LAUNCH(loss_grad_kernel<float>, dim3(G), dim3(kGradT), 0, st, a, J);
    else VAA_LAUNCH(loss_grad_kernel<uint16_t>, dim3(G), dim3(kGradT), 0, st, a, J);
    rc = check_launch("vaa_loss_fwd_bwd(grad)");
    return rc;
}

extern "C" int vaa_loss_fwd_bwd(const void* logits, int dtype, int layout, const int64_t* labels, int B, int S, int L, int V,
                                int mode, const float* params, float* scalars, int32_t* pred_tokens, void* glogits, void* ws,
                                size_t ws_bytes, void* stream) {
    return vaa_loss_fwd_bwd_ex(logits, dtype, layout, labels, B, S, L, V, mode, params, scalars, pred_tokens, nullptr, glogits, ws, ws_bytes, stream);
}

extern "C" size_t vaa_loss_rowmap_bytes(int B, int L) {
    if (B <= 0 || L <= 1) return 0;
    return 16 + (size_t)B * (size_t)(L - 1) * sizeof(vaa::RowMap);
}

extern "C" int vaa_loss_rowmap_build(const int64_t* labels, int B, int L, void* rowmap, size_t rowmap_bytes, void* stream) {
    using namespace vaa;
    if (!labels || !rowmap || B <= 0 || L <= 1) {
        set_error("vaa_loss_rowmap_build: bad arguments (B=%d L=%d)", B, L);
        return VAA_E_INVALID;
    }
    if (rowmap_bytes < vaa_loss_rowmap_bytes(B, L)) {
        set_error("vaa_loss_rowmap_build: buffer %zu B < required %zu B", rowmap_bytes, vaa_loss_rowmap_bytes(B, L));
        return VAA_E_WORKSPACE;
    }
    VAA_LAUNCH(loss_rowmap_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, B, L, (int*)rowmap);
    return check_launch("vaa_loss_rowmap_build");
}

extern "C" size_t vaa_loss_rows_ws_bytes(int R) {
    if (R <= 0) return 0;
    return (size_t)R * (4 * sizeof(vaa::PartStat) + sizeof(vaa::SliceStat));
}

namespace vaa {

// argument checks + RowsArgs shared by vaa_loss_rows_fwd_bwd / vaa_loss_rows_stats / vaa_step_epilogue
static int rows_args(const char* who, const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode, const float* params,
                     float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* grad, int grad_kind, void* ws, size_t ws_bytes, RowsArgs& a) {
    if (!rowmap || !params) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (R <= 0 || B <= 0 || L <= 1 || V < kA0 + kNA || (V % 8) != 0 || mode < 0 || mode > VAA_LOSS_CE ||
        (dtype != VAA_DTYPE_F32 && dtype != VAA_DTYPE_BF16) || (grad_kind != VAA_GRAD_FULL && grad_kind != VAA_GRAD_SLICE)) {
        set_error("%s: bad sizes/mode (R=%d B=%d L=%d V=%d mode=%d dtype=%d grad_kind=%d)", who, R, B, L, V, mode, dtype, grad_kind);
        return VAA_E_INVALID;
    }
    if (grad && grad_kind == VAA_GRAD_SLICE && (mode == VAA_LOSS_UADA || mode == VAA_LOSS_CE)) {
        set_error("%s: mode %d has a cross-entropy term, its gradient is not confined to the action slice", who, mode);
        return VAA_E_INVALID;
    }
    if ((long)R > (long)B * (L - 1)) {
        set_error("%s: R=%d exceeds the B*(L-1)=%ld label positions of the row map", who, R, (long)B * (L - 1));
        return VAA_E_INVALID;
    }
    if (V > 4 * kRowsTMax * 32) {
        set_error("%s: vocabulary %d exceeds the %d columns the row kernels keep in registers", who, V, 4 * kRowsTMax * 32);
        return VAA_E_UNSUPPORTED;
    }
    if (!ws || ws_bytes < vaa_loss_rows_ws_bytes(R)) {
        set_error("%s: workspace %zu B < required %zu B", who, ws_bytes, vaa_loss_rows_ws_bytes(R));
        return VAA_E_WORKSPACE;
    }
    a.logits = logits; a.rowmap = (const int*)rowmap; a.part = (PartStat*)ws; a.slice = (SliceStat*)((char*)ws + (size_t)R * 4 * sizeof(PartStat));
    a.grad = grad; a.scalars = scalars; a.pred_tokens = pred_tokens; a.pred_full = pred_full_tokens;
    a.R = R; a.B = B; a.L = L; a.V = V; a.mode = mode; a.split = rows_split(R, V); a.grad_slice = (grad_kind == VAA_GRAD_SLICE) ? 1 : 0;
    a.ldz = V; a.zcol0 = 0;
    a.w = params[0]; a.alpha = params[1]; a.beta = params[2]; a.scale = params[3];
    return VAA_OK;
}

// Hand-over words of the one-pass form, in the library's own device image (nothing is allocated, the caller's workspace keeps its
// "contents undefined" contract): per stream that uses it twelve 128-byte lines — line 0 the second-level arrival counter, lines 1..8 the
// first-level counters (blockIdx & 7), line 9 the two published words. The counters are re-armed by their last arrival, the published
// words carry a per-slot launch generation, so a launch leaves nothing to clean up.
constexpr int kBarSlots = 64, kBarWords = 32 * 12;
__device__ unsigned g_rows_bar[kBarSlots][kBarWords];

static unsigned* rows_bar_for(hipStream_t st, unsigned* gen) {
    static std::mutex mu;
    static hipStream_t owner[kBarSlots];
    static unsigned gens[kBarSlots];
    static int used = 0;
    static unsigned* base[16];  // the array's address on each device of the process
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (!base[dev] && hipGetSymbolAddress(reinterpret_cast<void**>(&base[dev]), HIP_SYMBOL(g_rows_bar)) != hipSuccess) {
        (void)hipGetLastError();
        base[dev] = nullptr;
        return nullptr;
    }
    int i = 0;
    while (i < used && owner[i] != st) ++i;
    if (i == used) {
        if (used == kBarSlots) return nullptr;  // more streams than slots: the caller falls back to the two-launch form
        owner[used] = st;
        gens[used++] = 0u;
    }
    if (++gens[i] == 0u) ++gens[i];  // never 0: the words start zeroed
    *gen = gens[i];
    return base[dev] + (size_t)kBarWords * i;
}

// The one-pass form waits on a grid-wide hand-over, so every workgroup of the launch has to be resident at once: the grid is admitted
// only when the occupancy the runtime reports for that instantiation (x the device's CU count) covers it; otherwise two launches.
template <typename K>
static bool rows_grid_resident(K kernel, int threads, long grid, int variant) {
    static std::atomic<long> slots[4][16];  // [kernel variant][device]: resident workgroups, queried once (0 = not yet, -1 = query failed)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
        (void)hipGetLastError();
        return false;
    }
    long have = slots[variant][dev].load(std::memory_order_relaxed);
    if (have == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) {
            (void)hipGetLastError();
            have = -1;
        } else {
            have = (long)cus * per_cu;
        }
        slots[variant][dev].store(have, std::memory_order_relaxed);
    }
    const long cus_per_cu = have;
    // half of what fits: a second launch of the same shape from ANOTHER process (the GPU shared by two ranks or tenants) still finds room,
    // so two waiting grids can never hold the slots each other's last workgroups need
    return cus_per_cu > 0 && 2 * grid <= cus_per_cu;
}

// Inside this process at most ONE stream has one-pass launches in flight: a request from another stream is admitted only once the owner
// stream has drained (hipStreamQuery), else it takes the two-launch form — two half-resident waiting grids would otherwise be possible.
bool rows_one_pass_stream_ok(hipStream_t st) {
    static std::mutex mu;
    static hipStream_t owner = nullptr;
    static bool owned = false;
    std::lock_guard<std::mutex> lk(mu);
    if (owned && owner != st) {
        const hipError_t q = hipStreamQuery(owner);
        if (q == hipErrorNotReady) return false;
        if (q != hipSuccess) (void)hipGetLastError();  // the owner stream no longer exists: nothing of it is in flight
    }
    owner = st;
    owned = true;
    return true;
}

constexpr bool kOnePassDefault = false;  // opt-in since round 4 (VAA_K3_ONE_PASS=1): 2.8 us per call do not pay for a residency assumption
static bool rows_one_pass_wanted() {  // VAA_K3_ONE_PASS=1 / 0 overrides the default
    const char* ev = getenv("VAA_K3_ONE_PASS");
    return ev && *ev ? (ev[0] != '0') : kOnePassDefault;
}

static bool rows_one_pass_fits(const RowsArgs& a, int dtype) {
    const long grid = (long)a.R * a.split;
    if (rows_threads(a.V) == 256)
        return dtype == VAA_DTYPE_F32 ? rows_grid_resident(rows_stats_kernel<float, 256, true>, 256, grid, 0)
                                      : rows_grid_resident(rows_stats_kernel<uint16_t, 256, true>, 256, grid, 1);
    return dtype == VAA_DTYPE_F32 ? rows_grid_resident(rows_stats_kernel<float, 512, true>, 512, grid, 2)
                                  : rows_grid_resident(rows_stats_kernel<uint16_t, 512, true>, 512, grid, 3);
}

static int launch_rows_stats(const RowsArgs& a, int dtype, hipStream_t st, const char* who, unsigned* bar = nullptr, unsigned gen = 0u) {
    unsigned* err = bar ? async_error_word() : nullptr;
    int polls = 1 << 22;
    if (bar) {  // test hook: VAA_K3_HANDOVER_POLLS=0 makes every waiting workgroup give up at once (the failure path's test)
        const char* ev = getenv("VAA_K3_HANDOVER_POLLS");
        if (ev && *ev) polls = atoi(ev);
    }
    const int nt = rows_threads(a.V);
    const dim3 gs((unsigned)(a.R * a.split));
    unsigned* nobar = nullptr;
    if (bar) {  // one pass: statistics, grid barrier, fold, full-row gradient from the registers
        if (dtype == VAA_DTYPE_F32) {
            if (nt == 256) VAA_LAUNCH((rows_stats_kernel<float, 256, true>), gs, dim3(256), 0, st, a, bar, gen, err, polls);
            else VAA_LAUNCH((rows_stats_kernel<float, 512, true>), gs, dim3(512), 0, st, a, bar, gen, err, polls);
        } else {
            if (nt == 256) VAA_LAUNCH((rows_stats_kernel<uint16_t, 256, true>), gs, dim3(256), 0, st, a, bar, gen, err, polls);
            else VAA_LAUNCH((rows_stats_kernel<uint16_t, 512, true>), gs, dim3(512), 0, st, a, bar, gen, err, polls);
        }
    } else if (dtype == VAA_DTYPE_F32) {
        if (nt == 256) VAA_LAUNCH((rows_stats_kernel<float, 256, false>), gs, dim3(256), 0, st, a, nobar, 0u, nobar, 0);
        else VAA_LAUNCH((rows_stats_kernel<float, 512, false>), gs, dim3(512), 0, st, a, nobar, 0u, nobar, 0);
    } else {
        if (nt == 256) VAA_LAUNCH((rows_stats_kernel<uint16_t, 256, false>), gs, dim3(256), 0, st, a, nobar, 0u, nobar, 0);
        else VAA_LAUNCH((rows_stats_kernel<uint16_t, 512, false>), gs, dim3(512), 0, st, a, nobar, 0u, nobar, 0);
    }
    return check_launch(who);
}

}  // namespace vaa

extern "C" int vaa_loss_rows_fwd_bwd(const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode,
                                     const float* params, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* grad,
                                     int grad_kind, void* ws, size_t ws_bytes, void* stream) {
    using namespace vaa;
    const char* who = "vaa_loss_rows_fwd_bwd";
    if (!logits || !scalars) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (R == 0 && B > 0 && L > 1) {  // nothing is labelled: every scalar is 0 (the reference's means over empty sets are not defined), no prediction
        hipStream_t st0 = (hipStream_t)stream;
        hipError_t e = hipMemsetAsync(scalars, 0, 8 * sizeof(float), st0);
        if (e == hipSuccess && pred_tokens) e = hipMemsetAsync(pred_tokens, 0xff, (size_t)B * (L - 1) * sizeof(int32_t), st0);
        if (e == hipSuccess && pred_full_tokens) e = hipMemsetAsync(pred_full_tokens, 0xff, (size_t)B * (L - 1) * sizeof(int32_t), st0);
        if (e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); return VAA_E_LAUNCH; }
        return VAA_OK;
    }
    RowsArgs a;
    int rc = rows_args(who, logits, dtype, rowmap, R, B, L, V, mode, params, scalars, pred_tokens, pred_full_tokens, grad, grad_kind, ws, ws_bytes, a);
    if (rc != VAA_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // Full-row gradients (UADA's 1/CE^2, CE) in ONE launch: the statistics pass keeps its logits in registers across a grid-wide hand-over
    // and writes the gradient from them, so every row is read once (16.4 MB instead of 25.4 MB at R' = 128) — bitwise the outputs of the
    // two launches below, 19.2 against 21.1 us per call in a stream at R' = 128 (tools/k3_onepass_check.py, profiles/r03_k3_onepass.txt).
    // Admitted when the grid takes at most HALF of the slots the runtime reports (residency is what a waiting grid relies on), outside
    // stream capture (the hand-over generation is a launch argument) and while no OTHER stream of this process has such launches in
    // flight; anything else takes the two launches. VAA_K3_ONE_PASS=0 turns it off.
    if (rows_one_pass_wanted() && grad && grad_kind == VAA_GRAD_FULL && (mode == VAA_LOSS_UADA || mode == VAA_LOSS_CE) && (long)R * a.split <= 1024 && R < 65536) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const hipError_t ce = hipStreamIsCapturing(st, &cs);
        if (ce != hipSuccess) (void)hipGetLastError();
        const bool capturing = ce != hipSuccess || cs != hipStreamCaptureStatusNone;
        unsigned gen = 0u;
        unsigned* bar = (capturing || !rows_one_pass_fits(a, dtype) || !rows_one_pass_stream_ok(st)) ? nullptr : rows_bar_for(st, &gen);
        if (bar) return launch_rows_stats(a, dtype, st, "vaa_loss_rows_fwd_bwd(one pass)", bar, gen);
    }
    rc = launch_rows_stats(a, dtype, st, "vaa_loss_rows_fwd_bwd(stats)");
    if (rc != VAA_OK) return rc;
    // the finishing pass: per (row, part) when a full-row gradient (or a zero fill) has to be written, else one workgroup per row
    // (UPA slice) or a single workgroup (UADA_DDP slice: only the scalars are left to do)
    const int nt = rows_threads(V);
    const bool full_rows = grad && grad_kind == VAA_GRAD_FULL;
    const int gsplit = full_rows ? a.split : 1;
    unsigned G = full_rows ? (unsigned)(R * gsplit) : ((grad && mode == VAA_LOSS_UPA) ? (unsigned)R : 1u);
    // CE full-row gradient: the fold moves to a workgroup of its own (rows_finish_kernel); VAA_K3_CE_FOLD_WG=0 keeps the all-fold form
    const char* ce_ev = getenv("VAA_K3_CE_FOLD_WG");
    const bool ce_fold_wg = !(ce_ev && ce_ev[0] == '0');
    const int fold_wg = (full_rows && mode == VAA_LOSS_CE && ce_fold_wg) ? (int)G : -1;
    if (fold_wg >= 0) ++G;
    if (dtype == VAA_DTYPE_F32) {
        if (nt == 256) VAA_LAUNCH((rows_finish_kernel<float, 256>), dim3(G), dim3(256), 0, st, a, gsplit, fold_wg);
        else VAA_LAUNCH((rows_finish_kernel<float, 512>), dim3(G), dim3(512), 0, st, a, gsplit, fold_wg);
    } else {
        if (nt == 256) VAA_LAUNCH((rows_finish_kernel<uint16_t, 256>), dim3(G), dim3(256), 0, st, a, gsplit, fold_wg);
        else VAA_LAUNCH((rows_finish_kernel<uint16_t, 512>), dim3(G), dim3(512), 0, st, a, gsplit, fold_wg);
    }
    return check_launch("vaa_loss_rows_fwd_bwd(finish)");
}

// The finishing pass behind vaa_head_loss_rows_stats (LM head fused with K3's statistics): the fold of the rows into the scalars, the
// prediction maps and — VAA_LOSS_UPA — the gradient slice, which needs the folded batch means (UPA.py:375-387). It is rows_finish_kernel,
// the second launch of vaa_loss_rows_fwd_bwd, reading the 256 action logits of a row from the buffer the head kernel left instead of
// from [R,V] logits: the same bits for the same logits.
extern "C" int vaa_head_loss_rows_finish(const void* rowmap, int R, int B, int L, int V, int mode, const float* params, void* loss_ws,
                                         size_t loss_ws_bytes, const void* head_ws, size_t head_ws_bytes, float* scalars, int32_t* pred_tokens,
                                         int32_t* pred_full_tokens, void* grad_slice, void* stream) {
    using namespace vaa;
    const char* who = "vaa_head_loss_rows_finish";
    if (!scalars || !head_ws) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (grad_slice && mode != VAA_LOSS_UPA) {
        set_error("%s: gradient slice asked for mode %d (VAA_LOSS_UADA_DDP: vaa_head_loss_rows_stats writes it; modes with a cross-entropy term need "
                  "the [R,V] logits: LM-head GEMM + vaa_loss_rows_fwd_bwd)", who, mode);
        return mode == VAA_LOSS_UADA_DDP ? VAA_E_INVALID : VAA_E_UNSUPPORTED;
    }
    if (R > 0 && head_ws_bytes < vaa_head_loss_ws_bytes(R, V)) {
        set_error("%s: head workspace %zu B < required %zu B", who, head_ws_bytes, vaa_head_loss_ws_bytes(R, V));
        return VAA_E_WORKSPACE;
    }
    RowsArgs a;
    const void* zs = (const char*)head_ws + head_ws_slice_offset(R, V);
    int rc = rows_args(who, zs, VAA_DTYPE_BF16, rowmap, R, B, L, V, mode, params, scalars, pred_tokens, pred_full_tokens, grad_slice, VAA_GRAD_SLICE,
                       loss_ws, loss_ws_bytes, a);
    if (rc != VAA_OK) return rc;
    a.ldz = kNA;
    a.zcol0 = kA0;
    hipStream_t st = (hipStream_t)stream;
    const unsigned G = grad_slice ? (unsigned)R : 1u;  // one workgroup per row writes its slice; the scalars alone take one workgroup
    if (rows_threads(V) == 256) VAA_LAUNCH((rows_finish_kernel<uint16_t, 256>), dim3(G), dim3(256), 0, st, a, 1, -1);
    else VAA_LAUNCH((rows_finish_kernel<uint16_t, 512>), dim3(G), dim3(512), 0, st, a, 1, -1);
    return check_launch(who);
}

// The statistics pass of vaa_loss_rows_fwd_bwd alone (UADA_DDP mode: it also writes the gradient slice, which needs nothing from other
// rows). The scalars and the prediction maps are then produced by vaa_step_epilogue from the same workspace.
extern "C" int vaa_loss_rows_stats(const void* logits, int dtype, const void* rowmap, int R, int B, int L, int V, int mode, const float* params,
                                   void* grad, int grad_kind, void* ws, size_t ws_bytes, void* stream) {
    using namespace vaa;
    const char* who = "vaa_loss_rows_stats";
    if (!logits) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (grad && mode != VAA_LOSS_UADA_DDP) {
        set_error("%s: only VAA_LOSS_UADA_DDP has a gradient that does not depend on the folded scalars (mode %d: use vaa_loss_rows_fwd_bwd)", who, mode);
        return VAA_E_INVALID;
    }
    RowsArgs a;
    int rc = rows_args(who, logits, dtype, rowmap, R, B, L, V, mode, params, nullptr, nullptr, nullptr, grad, grad_kind, ws, ws_bytes, a);
    if (rc != VAA_OK) return rc;
    return launch_rows_stats(a, dtype, (hipStream_t)stream, who);
}

namespace vaa {

static int step_epilogue_impl(const char* who, const float* partials, int nparts, int n, const void* rowmap, int R, int B, int L, int V, int mode,
                              const float* params, const void* loss_ws, size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens,
                              int32_t* pred_full_tokens, float* msg, const UpdArgs* upd, double* stat_part, void* stream) {
    if (!partials || !msg || !scalars || nparts <= 0 || n <= 0) {
        set_error("%s: bad arguments (nparts=%d n=%d)", who, nparts, n);
        return VAA_E_INVALID;
    }
    EpiArgs e = {};
    e.partials = partials; e.msg = msg; e.scalars_in = scalars; e.n = n; e.nparts = nparts; e.nred = (n + 63) / 64; e.fold = rowmap ? 1 : 0;
    e.fuse_update = upd ? 1 : 0;
    if (upd) e.upd = *upd;
    e.stat_part = stat_part;
    RowsArgs a = {};
    if (rowmap) {
        int rc = rows_args(who, nullptr, VAA_DTYPE_BF16, rowmap, R, B, L, V, mode, params, scalars, pred_tokens, pred_full_tokens, nullptr, VAA_GRAD_SLICE,
                           const_cast<void*>(loss_ws), loss_ws_bytes, a);
        if (rc != VAA_OK) return rc;
    }
    VAA_LAUNCH(step_epilogue_kernel, dim3((unsigned)(e.nred + 1)), dim3(256), 0, (hipStream_t)stream, e, a);
    return check_launch(who);
}

}  // namespace vaa

extern "C" int vaa_step_epilogue(const float* partials, int nparts, int n, const void* rowmap, int R, int B, int L, int V, int mode,
                                 const float* params, const void* loss_ws, size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens,
                                 int32_t* pred_full_tokens, float* msg, void* stream) {
    return vaa::step_epilogue_impl("vaa_step_epilogue", partials, nparts, n, rowmap, R, B, L, V, mode, params, loss_ws, loss_ws_bytes, scalars, pred_tokens,
                                   pred_full_tokens, msg, nullptr, nullptr, stream);
}

// The single-GPU step has no exchange between the gradient and the update: K4 (vaa_patch_update without L1 clip, grad_scale = 1) is applied by
// the epilogue on every gradient element as it is produced — same per-element arithmetic, same bits in patch / m / v. The logged statistics come
// back as per-block partial sums stat_part [ceil(n/64)][2] = {sum |g|, sum g} (fp64) for the caller to add.
extern "C" int vaa_step_epilogue_update(const float* partials, int nparts, int n, const void* rowmap, int R, int B, int L, int V, int mode,
                                        const float* params, const void* loss_ws, size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens,
                                        int32_t* pred_full_tokens, float* msg, float* patch, float* m, float* v, int opt_mode, float lr, float beta1,
                                        float beta2, float eps, int step, double* stat_part, void* stream) {
    using namespace vaa;
    const char* who = "vaa_step_epilogue_update";
    if (!patch || (opt_mode == VAA_OPT_ADAMW_HF && (!m || !v))) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if ((opt_mode != VAA_OPT_ADAMW_HF && opt_mode != VAA_OPT_PGD_SIGN) || (opt_mode == VAA_OPT_ADAMW_HF && step < 1)) {
        set_error("%s: bad optimiser mode/step (mode=%d step=%d)", who, opt_mode, step);
        return VAA_E_INVALID;
    }
    UpdArgs u = {};
    u.patch = patch; u.g = nullptr; u.m = m; u.v = v; u.stats = nullptr; u.n = n; u.mode = opt_mode;
    u.lr = lr; u.b1 = beta1; u.b2 = beta2; u.eps = eps; u.l1_clip = 0.0f; u.grad_scale = 1.0f;
    const double b1 = (double)beta1, b2 = (double)beta2;  // python-side doubles of the reference optimiser, narrowed exactly like vaa_patch_update
    u.one_m_b1 = (float)(1.0 - b1);
    u.one_m_b2 = (float)(1.0 - b2);
    u.step_size = (opt_mode == VAA_OPT_ADAMW_HF) ? (float)((double)lr * sqrt(1.0 - pow(b2, (double)step)) / (1.0 - pow(b1, (double)step))) : 0.0f;
    return step_epilogue_impl(who, partials, nparts, n, rowmap, R, B, L, V, mode, params, loss_ws, loss_ws_bytes, scalars, pred_tokens,
                              pred_full_tokens, msg, &u, stat_part, stream);
}
