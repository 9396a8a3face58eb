// vaa_rows_fold.h — the fold of K3's per-row statistics into the loss scalars (and what it is made of), shared by vaa_loss.hip (K3, the step
// epilogue) and vaa_head_slice.hip (K3s: the slice-only LM head that folds inside its own launch). One source, same bits.
#pragma once
#include <math.h>

#include "vaa_common.h"
#include "vaa_rows.h"

namespace vaa {

struct RowStat {  // one per LABELLED row, stored compactly at its row-major rank `rowidx` among labelled positions
    float lse;   // logsumexp over all V classes
    float zlab;  // logit of the label
    float alse;  // logsumexp over the 256 action classes
    float E;     // sum_a softmax_a * (a+1), in [1,256]
    int pred;    // 31744 + argmax over the action slice
    int pos;     // position p = b*(L-1)+k
    int lab;     // the label (token id)
    int ord;     // rank of this position among its sample's labelled positions (0 = first)
    int predf;   // argmax over ALL V classes (lowest index on ties, torch.argmax)
};

__device__ __forceinline__ double bin_center(int tok) {  // ActionTokenizer.decode_token_ids_to_actions (action_tokenizer.py:49-68)
    int d = 32000 - tok - 1;
    d = d < 0 ? 0 : (d > 254 ? 254 : d);
    return -1.0 + (2.0 * d + 1.0) / 255.0;
}

// block-wide sums of NV doubles at once (fixed order: lanes by xor-shuffle, then waves 0..15) -> deterministic
template <int NV, int NT = 1024>
__device__ __forceinline__ void block_sums(double (&v)[NV], double (*sh)[NV]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) sh[wv][q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double t = 0.0;
        for (int w = 0; w < NT / 64; ++w) t += sh[w][q];
        v[q] = t;
    }
}

// UPA per-sample terms from the E / label of a sample's first three labelled rows (UPA.py:375-384)
struct Upa3 {
    double e[3], l[3];
    __device__ void set(int q, const RowStat& s) { e[q] = ((double)s.E - 1.0) / 255.0; l[q] = ((double)(s.lab - 31743) - 1.0) / 255.0; }
    __device__ void terms(double& cosp1, double& nd) const {
        const double dot = e[0] * l[0] + e[1] * l[1] + e[2] * l[2];
        const double ne = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], nl = l[0] * l[0] + l[1] * l[1] + l[2] * l[2];
        const double d0 = e[0] - l[0], d1 = e[1] - l[1], d2 = e[2] - l[2];
        cosp1 = dot / (fmax(sqrt(ne), 1e-8) * fmax(sqrt(nl), 1e-8)) + 1.0;  // F.cosine_similarity + 1 (UPA.py:382-383)
        nd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    }
    __device__ double dE(int q, double alpha, double beta, double aux1, int B) const {  // d total / d e[q]
        const double dot = e[0] * l[0] + e[1] * l[1] + e[2] * l[2];
        const double ne = e[0] * e[0] + e[1] * e[1] + e[2] * e[2], nl = l[0] * l[0] + l[1] * l[1] + l[2] * l[2];
        const double d0 = e[0] - l[0], d1 = e[1] - l[1], d2 = e[2] - l[2];
        const double sne = fmax(sqrt(ne), 1e-8), snl = fmax(sqrt(nl), 1e-8), nd = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        const double eq = q == 0 ? e[0] : (q == 1 ? e[1] : e[2]), lq = q == 0 ? l[0] : (q == 1 ? l[1] : l[2]);
        const double dcos = lq / (sne * snl) - dot * eq / (sne * sne * sne * snl);
        const double dn = nd > 0 ? (eq - lq) / nd : 0.0;
        return alpha * dcos / B - beta * aux1 * aux1 * dn / B;
    }
};

struct RowsArgs {
    const void* logits;
    const int* rowmap;  // hdr[4] + RowMap[R]
    PartStat* part;     // [R][split]
    SliceStat* slice;   // [R]
    void* grad;
    float* scalars;
    int32_t* pred_tokens;
    int32_t* pred_full;
    int R, B, L, V, mode, split, grad_slice;
    int ldz, zcol0;     // logits row stride (elements) and the vocabulary column of its first element: V, 0 — or 256, kA0 when `logits` is the
                        // action-column buffer vaa_head_loss_rows_stats left (vaa_head_loss_rows_finish: only the slice is ever read)
    float w, alpha, beta, scale;
};

struct FoldOut {
    double nrow, nact, CE, MSE, UAD, total, aux0, aux1, dce;
    int Rn;
};
// one-pass hand-over (rows_stats_kernel<.., true>): the folding workgroup leaves every row's log-sum-exp beside its slice statistics
// (SliceStat::pad) and publishes what every gradient needs as two SELF-VALIDATING 64-bit words {generation, payload} — a waiting workgroup
// polls exactly those two words and needs no ordering between them — BEFORE it writes the scalars and prediction maps nobody waits for
struct Handover {
    unsigned long long* words;  // [0] = {gen, bits of kce = dce / nrow as fp32}, [1] = {gen, nact | Rn << 16}
    unsigned gen;
};
template <int kRowsT, bool COH = false>
__device__ __forceinline__ FoldOut rows_fold(const RowsArgs& a, bool publish, double (*sh)[7], Handover ho = Handover{nullptr, 0u});

// statistics words another workgroup of the SAME launch wrote: agent-scope atomics go past the per-XCD L2 (COH); plain accesses otherwise
template <bool COH, typename S>
__device__ __forceinline__ S stat_load(const S* p) {
    static_assert(sizeof(S) == 16, "four words");
    if (!COH) return *p;
    unsigned w[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) w[z] = __hip_atomic_load(reinterpret_cast<const unsigned*>(p) + z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    S v;
    __builtin_memcpy(&v, w, 16);
    return v;
}
template <bool COH, typename S>
__device__ __forceinline__ void stat_store(S* p, const S& v) {
    static_assert(sizeof(S) == 16, "four words");
    if (!COH) { *p = v; return; }
    unsigned w[4];
    __builtin_memcpy(w, &v, 16);
#pragma unroll
    for (int z = 0; z < 4; ++z) __hip_atomic_store(reinterpret_cast<unsigned*>(p) + z, w[z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Folds the R compact row statistics into the loss scalars in a fixed order (thread t takes rows t, t + kRowsT, ...; block_sums), and — for
// the publishing workgroup — writes scalars[8] and the two prediction maps. Shared by rows_finish_kernel and the step epilogue.
template <int kRowsT, bool COH>
__device__ __forceinline__ FoldOut rows_fold(const RowsArgs& a, bool publish, double (*sh)[7], Handover ho) {
    const int tid = threadIdx.x;
    const RowMap* rm = reinterpret_cast<const RowMap*>(a.rowmap + 4);
    const int Rdev = a.rowmap[0];
    const int Rn = min(a.R, Rdev);  // rows both the caller and the map know: a mismatch publishes NaN and never leaves the map
    auto row_lse = [&](int rr, float& zlab, int& amax) {  // combine the parts of row rr
        float M = -INFINITY, tot = 0.0f, best = -INFINITY;
        zlab = -INFINITY;
        amax = 0x7fffffff;
        constexpr int kHeld = 8;
        if (COH && a.split <= kHeld) {  // coherent loads are not merged by the compiler: fetch every part ONCE, all requests in flight together
            PartStat ps[kHeld];
#pragma unroll
            for (int q = 0; q < kHeld; ++q)
                if (q < a.split) ps[q] = stat_load<COH>(&a.part[(size_t)rr * a.split + q]);
#pragma unroll
            for (int q = 0; q < kHeld; ++q)
                if (q < a.split) M = fmaxf(M, ps[q].m);
#pragma unroll
            for (int q = 0; q < kHeld; ++q)
                if (q < a.split) {
                    tot += ps[q].s * expf(ps[q].m - M);
                    zlab = fmaxf(zlab, ps[q].zlab);
                    if (ps[q].m > best || (ps[q].m == best && ps[q].amax < amax)) { best = ps[q].m; amax = ps[q].amax; }
                }
            if (M == -INFINITY) { zlab = 0.0f; amax = -1; return 0.0f; }  // every part neutral: a slice-only step (see below)
            return M + logf(tot);
        }
        for (int q = 0; q < a.split; ++q) M = fmaxf(M, stat_load<COH>(&a.part[(size_t)rr * a.split + q]).m);
        for (int q = 0; q < a.split; ++q) {
            const PartStat p = stat_load<COH>(&a.part[(size_t)rr * a.split + q]);
            tot += p.s * expf(p.m - M);
            zlab = fmaxf(zlab, p.zlab);
            if (p.m > best || (p.m == best && p.amax < amax)) { best = p.m; amax = p.amax; }
        }
        // every part neutral {m = -inf, s = 0}: the row has NO full-vocabulary statistics — a slice-only step (vaa_head_slice_fwd_bwd: the loop
        // reads CE / the full argmax on 1 of innerLoop steps, UADA_ddp.py:214-221, and never in UPA's reverse-direction mode, UPA.py:145-186).
        // The row then adds 0 to CE and -1 to pred_full; real logits never have a maximum of -inf
        if (M == -INFINITY) { zlab = 0.0f; amax = -1; return 0.0f; }
        return M + logf(tot);
    };
    const bool handing = COH && ho.words != nullptr;
    auto clear_maps = [&]() {
        const int P0 = a.B * (a.L - 1);
        if (a.pred_tokens) for (int q = tid; q < P0; q += kRowsT) a.pred_tokens[q] = -1;
        if (a.pred_full) for (int q = tid; q < P0; q += kRowsT) a.pred_full[q] = -1;
    };
    // the publishing workgroup clears the prediction maps NOW, under the statistics' load latency (the barriers of the block reduction below
    // order these stores before the per-row stores that follow it) — unless a grid is waiting for this fold: then nothing is queued in front
    // of the statistics loads and the maps are cleared after the hand-over
    if (publish && !handing) clear_maps();
    int first_am = 0;  // full-vocabulary argmax of this thread's first row (rr = tid), kept for the publication
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};  // ce, mse, uad, nrow, nact, upa sum(cos+1), upa sum ||e'-l'||
    for (int rr = tid; rr < Rn; rr += kRowsT) {
        const RowMap m = rm[rr];
        float zl;
        int am;
        SliceStat ss_early = {0.f, 0.f, 0, 0};
        if (handing) ss_early = stat_load<COH>(&a.slice[rr]);  // requested together with the parts: one round trip for the whole row
        const float lse = row_lse(rr, zl, am);
        if (handing)  // the row's workgroups take their log-sum-exp from here instead of combining the parts again
            __hip_atomic_store(reinterpret_cast<unsigned*>(&a.slice[rr].pad), __float_as_uint(lse), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rr == tid) first_am = am;
        acc[3] += 1.0;
        acc[0] += (double)lse - (double)zl;
        if (m.lab > 2) {
            const SliceStat ss = handing ? ss_early : stat_load<COH>(&a.slice[rr]);
            acc[4] += 1.0;
            const double q = (double)ss.E / 256.0, t = (m.lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10: 1/256 -> 0)
            acc[1] += (q - t) * (q - t);
            const double ag = bin_center(m.lab), ap = bin_center(ss.pred);  // cal_UAD, UADA.py:408-418
            acc[2] += fabs(ap - ag) / (ag > 0 ? fabs(ag + 1.0) : fabs(ag - 1.0));
        }
        if (a.mode == VAA_LOSS_UPA && m.ord == 0 && rr + 2 < Rn) {  // first three labelled rows of a sample are consecutive ranks
            Upa3 u;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                RowStat t;
                t.E = stat_load<COH>(&a.slice[rr + q]).E;
                t.lab = rm[rr + q].lab;
                u.set(q, t);
            }
            double c1, nd;
            u.terms(c1, nd);
            acc[5] += c1;
            acc[6] += nd;
        }
    }
    block_sums<7, kRowsT>(acc, sh);
    FoldOut f;
    f.Rn = Rn;
    f.nrow = acc[3]; f.nact = acc[4];
    f.CE = f.nrow > 0 ? acc[0] / f.nrow : 0.0;
    f.MSE = f.nact > 0 ? (double)a.w * a.w * acc[1] / f.nact : 0.0;
    f.UAD = f.nact > 0 ? acc[2] / f.nact : 0.0;
    f.total = 0.0; f.aux0 = 0.0; f.aux1 = 0.0; f.dce = 0.0;
    if (a.mode == VAA_LOSS_UPA) {
        f.aux0 = acc[5] / a.B;
        f.aux1 = 1.0 / (acc[6] / a.B + 1e-3);  // UPA.py:384
        f.total = (double)a.alpha * f.aux0 + (double)a.beta * f.aux1;
    } else if (a.mode == VAA_LOSS_UADA) { f.total = f.MSE + 1.0 / f.CE; f.dce = -1.0 / (f.CE * f.CE); }  // UADA.py:147
    else if (a.mode == VAA_LOSS_UADA_DDP) { f.total = f.MSE; }                                           // UADA_ddp.py:203-206
    else { f.total = (double)a.scale * f.CE; f.dce = (double)a.scale; }                                 // TMA.py:148
    if (handing) {  // the waiting workgroups need {kce, nact, Rn}: hand over now, publish afterwards
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's log-sum-exp words have COMPLETED before the publication below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) {
            const float kce = f.nrow > 0 ? (float)(f.dce / f.nrow) : 0.0f;
            const unsigned long long g = (unsigned long long)ho.gen << 32;
            __hip_atomic_store(&ho.words[0], g | __float_as_uint(kce), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ho.words[1], g | ((unsigned)f.nact & 0xffffu) | ((unsigned)f.Rn << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (publish && handing) {
        clear_maps();
        __syncthreads();
    }
    if (publish) {
        if (tid == 0) {
            const bool ok = Rdev == a.R;  // the caller's row count must be the row map's
            a.scalars[0] = ok ? (float)f.total : __uint_as_float(0x7fc00000u);
            a.scalars[1] = (float)f.CE; a.scalars[2] = (float)f.MSE; a.scalars[3] = (float)f.aux0;
            a.scalars[4] = (float)f.aux1; a.scalars[5] = (float)f.nrow; a.scalars[6] = (float)f.nact; a.scalars[7] = (float)f.UAD;
        }
        for (int rr = tid; rr < Rn; rr += kRowsT) {
            const RowMap m = rm[rr];
            const int pos = m.b * (a.L - 1) + m.k;
            if ((unsigned)pos >= (unsigned)(a.B * (a.L - 1))) continue;  // a map built for other sizes than the caller states
            if (a.pred_tokens && m.lab > 2) a.pred_tokens[pos] = stat_load<COH>(&a.slice[rr]).pred;
            if (a.pred_full) {
                int am = first_am;
                if (rr != tid) {  // more rows than threads: combine the parts again
                    float zl;
                    row_lse(rr, zl, am);
                }
                a.pred_full[pos] = am;
            }
        }
    }
    return f;
}

// At most ONE stream of the process has grids in flight whose workgroups wait for each other (vaa_loss.hip's one-pass K3, vaa_head_slice.hip's K3s):
// a request from another stream is admitted only once the owner stream has drained — two half-resident waiting grids would otherwise be possible.
bool rows_one_pass_stream_ok(hipStream_t st);

}  // namespace vaa
