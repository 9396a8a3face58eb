// vaa_head_slice.hip — K3s: the LM head on the labelled rows restricted to the 256 ACTION columns, fused with the slice statistics, the loss
// gradient and the head's backward. ONE launch per inner step for the modes whose loss lives in the action columns (UADA_DDP, UPA).
//
// What a step of those loops consumes (UADA_ddp.py:99-124,196-221; UPA.py:139-186,367-387): the soft-argmax / MSE (or cos / dist) statistics of
// logits[:, 31744:32000] and their gradient. The full-vocabulary CE and argmax are read on the LAST inner step of an outer iteration only
// (UADA_ddp.py:214-221 logs `celoss` once per outer iteration) and never in UPA's reverse-direction mode — so the 263 MB weight stream of
// vaa_head.hip (K3h) runs on those steps only, and every step runs this kernel: [R' <= 128, D] x W[31744:32000]^T, 2.1 MB of weights.
//
//   grid = ceil(R'/16) row blocks x 32 workgroups of 8 waves (8-column form, the default; 16 workgroups in the 16-column form); workgroup (rb, j):
//     phase 1  logits tile [16 rows of block rb] x [8 action columns 8 j ..] (the 16 x 16 MFMA tile carries its eight columns twice): H and W rows go
//              global -> LDS by LDS-DMA in full 128-byte lines (8 rows x 128 B per instruction, the slot image and XOR placement of
//              head_stats_kernel), a ring of groups of four k-chunks with EXACT request counts, counted vmcnt waits and ONE raw s_barrier per group;
//              wave 0 keeps two half groups of fragments — the reads of one half travel while the four dependent MFMAs of the other run — and runs
//              the single mfma_f32_16x16x32_bf16 accumulator chain over K in the k-chunk order of the K3h workgroup that owns these columns (start
//              chunk (5 w) mod D/64): the SAME instruction sequence per output element, so the bf16-rounded logits are bit for bit K3h's.
//              What bounds the kernel is what ONE CU has to fetch (its fetch path moves 22-29 B/clk): 128 KB of hidden rows + 8 KB per action
//              column + (phase 3) 512 B per column of dH — 256 KB per CU on 256 CUs in this form, 384 KB per CU on 128 CUs in the 16-column form.
//     hand-over  the tile goes to a scratch of 64-bit words {launch tag : 32 | logit 2q+1 : 16 | logit 2q : 16} by agent-scope stores; a consumer
//              polls the very words it is going to use — two memory round trips. The grid is <= 256 workgroups of 72 KB LDS and 122 VGPRs (two per
//              CU) and admitted only when the device keeps twice that resident; otherwise the same kernel runs as two launches (phase 1, then
//              phases 2 + 3).
//     phase 2  every workgroup of row block rb reads the block's 16 x 256 logits and recomputes — with the arithmetic of head_finish_kernel /
//              rows_stats_kernel, one row per HALF wave — {alse, E, argmax} and the gradient slice g = kE p_a ((a+1) - E) (UADA_ddp.py:99-114;
//              UPA: kE from the batch means, UPA.py:375-387, all rows folded in the fixed order of rows_fold) as a bf16 [16,256] MFMA operand in
//              LDS. Workgroups j = 0 leave the SliceStats (and NEUTRAL full-vocabulary parts) in K3's workspace layout: vaa_step_epilogue folds
//              them as ever; or workgroup 0 folds and publishes scalars + prediction maps itself.
//     phase 3  dH[16 rows, 128 j .. 128 j + 128) = g [16,256] x W[31744:32000, those columns]: the B fragments come from a [D,256] transposed copy
//              of the slice (vaa_head_slice_pack, once per weight), requested in quarters BETWEEN the stages of phase 2 (a burst of all of them
//              stalls the statistics behind the CU's fetch rate; all of them in front of the poll: no better, 10.8 against 9.7 us at 16 rows).
//   Bytes per launch at R' = 128, D = 4096: H 1.05 MB + W slice 2.10 MB + transposed slice 2.10 MB read, dH 1.05 MB written = 6.3 MB
//   (K3h + finish + the 256-column GEMM: 263.7 + 1.05 + ~4.3 MB). 9.5 us per dispatch warm (16-column form: 10.7; round 6's first form 12.1),
//   12.6-12.7 us in the bs=64 step (14.1-14.2; 15.6-15.7): a latency chain (profiles/r06_k3s_stamps.txt).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "vaa_common.h"
#include "vaa_rows.h"
#include "vaa_rows_fold.h"

namespace vaa {

typedef short v8s_s __attribute__((ext_vector_type(8)));
typedef float v4f_s __attribute__((ext_vector_type(4)));
typedef float v2f_s __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf_s __attribute__((ext_vector_type(2)));

// {bf16(hi) : 16 | bf16(lo) : 16} in ONE instruction (v_cvt_pk_bf16_f32, round to nearest even): the bits f32_to_bf16_bits computes in seven for every
// value that is not a NaN (a NaN stays a NaN) — on this kernel's critical path (logits -> tagged words) and in its VALU-bound part (the gradient tile)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const v2f_s v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2bf_s));
}

constexpr int kST = 512;                    // 8 waves: 2 half waves x 8 = the 16 rows of a block in phase 2, 16 output tiles of 16 columns in phase 3
constexpr int kSK = 64;                     // k-chunk: one 128-byte line per row (head_stats_kernel's)
constexpr int kSGrp = 4;                    // k-chunks per group: one barrier, sixteen fragment reads, eight MFMAs
constexpr int kSGroups = 16;                // D = 4096: 64 k-chunks
#ifndef VAA_SLICE_RING
#define VAA_SLICE_RING 8
#endif
constexpr int kSRing = VAA_SLICE_RING;      // groups of 16 KB in LDS: kSRing - 1 in flight (two 1 KB instructions per wave and group)
#ifndef VAA_SLICE_RING8
#define VAA_SLICE_RING8 6
#endif
constexpr int kSRing8 = VAA_SLICE_RING8;    // ... of the 8-column form (groups of 12 KB): 72 KB, so that TWO workgroups fit a CU (its grid is up to 256)
constexpr int kSDMax = 4096;                // D covered: workgroups per row block x columns of dH per workgroup
constexpr int kSGS = kNA + 8;               // row stride (bf16) of the gradient / output tiles in LDS
constexpr int kSRowsMax = 128;


struct SliceHeadArgs {
    const uint16_t* h;   // [R, D] bf16 hidden rows (final norm applied)
    const uint16_t* w;   // [V, D] bf16 LM-head weight (rows 31744..31999 are read)
    const uint16_t* wt;  // [D, 256] bf16: wt[d][a] = w[31744 + a][d]
    unsigned long long* zs;  // [ceil16(R)][128] self-validating words {launch tag : 32 | logit 2q+1 : 16 | logit 2q : 16} (workspace)
    uint16_t* gs;        // [R][256] bf16 gradient slice or nullptr (tests)
    uint16_t* dh;        // [R][D] bf16 or nullptr (forward only)
    RowsArgs ra;         // row map, K3 workspace (part / slice), scalars, prediction maps, sizes, mode, params
    unsigned tag;        // this launch's tag (process-unique, never a plausible stale word)
    unsigned* err_word;
    int max_polls;
    int D, phases, publish;  // phases: bit 0 = logits, bit 1 = statistics / gradient / dH; publish: workgroup 0 folds scalars + prediction maps
};

// the slice statistics of NR rows, each held by a HALF wave (32 lanes x 8 logits), the rows' dependent shuffle chains side by side: the arithmetic of
// head_finish_kernel / rows_stats_kernel — whose 64-lane butterflies start at offset 32 against neutral upper lanes (-inf, 0), and whose separate
// wave_max of the lane maxima IS the value the argmax butterfly ends with (max is exact) — hence the same bits
// lane ^ O within a half wave: DPP quad permutes (O = 1, 2), a DPP row rotate (O = 8) and ds_swizzle's bit-mask mode (O = 4, 16; it works on 32-lane halves) — no
// address register and a shorter round trip than __shfl_xor's ds_bpermute; pure data movement, the same bits
template <int O>
__device__ __forceinline__ int xor_lane_i(int v) {
    if (O == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    if (O == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    if (O == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false); // row_ror:8 — within a row of 16 lanes, lane + 8 mod 16 IS lane ^ 8
    return __builtin_amdgcn_ds_swizzle(v, (O << 10) | 0x1F);                       // {xor = O, or = 0, and = 0x1f}
}
template <int O>
__device__ __forceinline__ float xor_lane(float v) { return __int_as_float(xor_lane_i<O>(__float_as_int(v))); }

struct NoHook {
    template <typename T>
    __device__ __forceinline__ void operator()(T) const {}
};
// `between(stage)` (stage = std::integral_constant 1, 2, 3) runs behind the argmax butterflies, the exponentials and the sum butterflies, fenced for the
// scheduler: the caller trickles independent memory requests in between instead of stalling the chain behind a burst of them
template <int NR, typename F = NoHook>
__device__ __forceinline__ void slice_stats_half(const float (&x)[NR][8], int hl, float (&alse)[NR], float (&E)[NR], int (&pred)[NR], F between = F()) {
    float bestv[NR], es[NR], ew[NR];
    int besti[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {  // first maximum of the lane's eight logits, value and index carried together (an index into x[] costs a select chain per step)
        float bv = x[n][0];
        int bi = 0;
#pragma unroll
        for (int e = 1; e < 8; ++e) {
            const bool gt = x[n][e] > bv;
            bv = gt ? x[n][e] : bv;
            bi = gt ? e : bi;
        }
        bestv[n] = bv;
        besti[n] = hl * 8 + bi;
    }
#define K3S_ARGMAX_STEP(O)                                                                                   \
    _Pragma("unroll") for (int n = 0; n < NR; ++n) {                                                        \
        const float ov = xor_lane<O>(bestv[n]);                                                             \
        const int oi = xor_lane_i<O>(besti[n]);                                                             \
        if (ov > bestv[n] || (ov == bestv[n] && oi < besti[n])) { bestv[n] = ov; besti[n] = oi; }           \
    }
    K3S_ARGMAX_STEP(16) K3S_ARGMAX_STEP(8) K3S_ARGMAX_STEP(4) K3S_ARGMAX_STEP(2) K3S_ARGMAX_STEP(1)
#undef K3S_ARGMAX_STEP
    __builtin_amdgcn_sched_barrier(0);
    between(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        es[n] = 0.0f;
        ew[n] = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ex = expf(x[n][e] - bestv[n]);
            es[n] += ex;
            ew[n] += ex * (float)(hl * 8 + e + 1);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    between(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
#define K3S_SUM_STEP(O)                                              \
    _Pragma("unroll") for (int n = 0; n < NR; ++n) {                    \
        es[n] += xor_lane<O>(es[n]);                                    \
        ew[n] += xor_lane<O>(ew[n]);                                    \
    }
    K3S_SUM_STEP(16) K3S_SUM_STEP(8) K3S_SUM_STEP(4) K3S_SUM_STEP(2) K3S_SUM_STEP(1)
#undef K3S_SUM_STEP
    __builtin_amdgcn_sched_barrier(0);
    between(std::integral_constant<int, 3>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        E[n] = ew[n] / es[n];
        alse[n] = bestv[n] + logf(es[n]);
        pred[n] = kA0 + besti[n];
    }
}

// The hand-over needs no counters: every 64-bit word of the logits scratch carries the launch's tag beside its two logits, so a consumer polls the
// very words it is going to use (agent-scope loads go past the per-XCD L2) — two memory round trips between the last MFMA of a producer and the
// first exp of a consumer (an arrival count + a published flag + the data costs five: 26.7 against 22.1 us for two launches at R' = 128).
__device__ __forceinline__ void decode_logits8(const unsigned long long (&wq)[4], float (&x)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { x[2 * q] = bf16_bits_to_f32((unsigned)wq[q] & 0xffffu); x[2 * q + 1] = bf16_bits_to_f32(((unsigned)wq[q] >> 16) & 0xffffu); }
}

// s_waitcnt vmcnt(n) for an n the unrolled loop knows at compile time
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// timing probe (build with -DVAA_K3S_TIMING; tools/probe/k3s_stamps.py): thread 0 of every workgroup leaves wall-clock stamps (10 ns units) behind the
// first 512 KB of the logits scratch — where the kernel's time goes, phase by phase
#ifdef VAA_K3S_TIMING
#define K3S_STAMP(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.zs) + (512 << 10))[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define K3S_STAMP(i) do { } while (0)
#endif

// CT = action columns per workgroup in phase 1: 16 (16 workgroups per row block) or 8 (32 workgroups per row block: the 16 x 16 MFMA tile carries
// its eight columns twice; what bounds this kernel is what ONE CU has to fetch — 128 KB of hidden rows + CT x 8 KB of weight rows + 4096 / (256 / CT)
// x 512 B of the transposed slice — so half the columns per workgroup on twice the CUs: 256 instead of 384 KB per CU)
template <int NRB, int CT>
__global__ __launch_bounds__(kST) __attribute__((amdgpu_waves_per_eu(CT == 8 ? 4 : 2, CT == 8 ? 4 : 2))) void head_slice_kernel(SliceHeadArgs a) {
    static_assert(CT == 16 || (CT == 8 && NRB == 1), "row-block groups exist in the 16-column form only");
    constexpr int kJ = kNA / CT;            // workgroups per row block
    constexpr int kRS = 16 + CT;            // rows of a chunk image: 16 hidden + CT weight rows
    [[maybe_unused]] constexpr int kSSlot = kRS * kSK;       // elements per chunk image
    [[maybe_unused]] constexpr int kOct = kRS / 8;           // LDS-DMA instructions (8 rows x 128 B) per chunk image
    [[maybe_unused]] constexpr int kRing = CT == 16 ? kSRing : kSRing8;
    extern __shared__ __align__(16) uint16_t smem[];  // phase 1: the ring [kRing][4 chunks][kRS rows][64]; afterwards the gradient and output tiles
    __shared__ float sAlse[kSRowsMax], sE[kSRowsMax];
    __shared__ double shf[kST / 64][7];
    __shared__ int bar_ok;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int rb = blockIdx.x / kJ, j = blockIdx.x % kJ;
    const int R = a.ra.R, D = a.D;
    const int nchunks = D / kSK;
    // phases 2 and 3 work on a GROUP of NRB consecutive row blocks: workgroup (rb, j) of group rg = rb / NRB computes the gradient rows of all
    // 16 NRB rows and the dH tile [those rows] x [256 / NRB columns of D]: the B operand of phase 3 is 128 KB / NRB per workgroup instead of 128 KB
    // (a CU's fetch path moves ~56 GB/s: the 128 KB stalled the statistics behind their issue for 2 us and skewed the waves by 1.8 us at NRB = 1)
    constexpr int kDW = kSDMax / kJ / NRB;  // columns of D per workgroup
    constexpr int kOS = kDW + 8;            // row stride (bf16) of the output tile
    uint16_t* gt = smem;                    // gradient tile [16 NRB][kSGS] ...
    uint16_t* ot = gt + 16 * NRB * kSGS;    // ... and output tile [16 NRB][kOS]
    const int rg = rb / NRB, mb = rb % NRB; // group, member
    const int hl = lane & 31, hw = lane >> 5;
    K3S_STAMP(0);
    // what phase 2 will want from global memory is requested now: the row map entry of this half wave's own row and the counts
    const RowMap* rm = reinterpret_cast<const RowMap*>(a.ra.rowmap + 4);
    const int Rdev = a.ra.rowmap[0], Rn = min(R, Rdev), nact = a.ra.rowmap[1];
    const int own_lr = 2 * wv + hw;  // 8 waves x 2 half waves = the 16 rows of a block; this half wave takes row own_lr of each block of the group
    RowMap me[NRB];
#pragma unroll
    for (int u = 0; u < NRB; ++u) {
        const int row = (rg * NRB + u) * 16 + own_lr;
        me[u] = row < Rn ? rm[row] : RowMap{0, 0, -1, 0};
    }

    if (a.phases & 1) {
        // ---- phase 1: 16 x 16 logits over all of K, in the k-chunk order of the K3h workgroup that owns columns 16 j .. (vaa_head.hip) ----
        const int k3h_wg = kA0 / kHeadCols + (j * CT) / kHeadCols;
        const int kstart = (int)(((unsigned)k3h_wg * 5u) % (unsigned)nchunks);
        auto kchunk = [&](int ch) { const int cc = ch + kstart; return cc >= nchunks ? cc - nchunks : cc; };
        auto lds_off = [](int row, int piece) { return row * kSK + ((piece ^ (row & 7)) << 3); };
        v4f_s acc = (v4f_s){0.f, 0.f, 0.f, 0.f};
        if (nchunks == kSGroups * kSGrp) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass does not know the LDS-DMA builtin)
            // D = 4096: sixteen groups of four 64-wide k-chunks. A chunk of the 16 hidden + 16 weight rows is 4 KB = four LDS-DMA instructions
            // (8 rows x 128 B: full lines, no staging registers — three register-staged forms ended in scratch memory or drained loads, 20 us
            // for the phase); the eight waves take the two chunk parities x four row octets, so a wave requests TWO instructions per group. A
            // ring of kSRing groups: group g + kSRing - 1 is requested in step g into the slots group g - 1 was read from, EXACT request counts
            // (the waits count down over the last groups: spare requests behind the end cost 2.6 us), counted vmcnt waits + ONE raw s_barrier
            // per group. Wave 0 reads a group's sixteen fragments, then runs its eight dependent MFMAs back to back (a read -> wait -> MFMA
            // per k-step: 9.4 against 7.0 us for the phase). The fetch path of a CU moves 22-29 B/clk of L2 hits with eight waves issuing
            // (tools/probe/cu_fetch_probe.hip): ~4.5 us for the workgroup's 256 KB is this phase's floor.
            const int wvu = __builtin_amdgcn_readfirstlane(wv);  // the DMA destination (M0) is per wave
            // (8-column form: three instructions per chunk image, waves 0 .. 5 request, waves 6 and 7 only keep the barriers)
            const int cw = wvu / kOct, r8 = (wvu % kOct) * 8;    // this wave's chunk parity within a pair and its row octet of the slot
            const bool dma = wvu < 2 * kOct;
            const int lrow = lane >> 3, lpiece = (lane & 7) ^ lrow;
            const uint16_t* src = r8 < 16 ? a.h + (size_t)min(rb * 16 + r8 + lrow, R - 1) * D + lpiece * 8
                                          : a.w + (size_t)(kA0 + j * CT + (r8 - 16) + lrow) * D + lpiece * 8;
            auto request_group = [&](int gi) {
                uint16_t* grp = smem + (size_t)(gi % kRing) * (kSGrp * kSSlot);
                if (dma) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = 2 * h + cw;
                        auto* dst = (__attribute__((address_space(3))) void*)(grp + q * kSSlot + r8 * kSK);
                        __builtin_amdgcn_global_load_lds(src + kchunk(gi * kSGrp + q) * kSK, dst, 16, 0, 0);
                    }
                }
            };
#pragma unroll
            for (int gi = 0; gi < kRing - 1; ++gi) request_group(gi);
            // wave 0 keeps TWO half groups of fragments (two k-chunks = eight 16-byte reads per operand pair each): the reads of one half travel
            // while the four dependent MFMAs of the other run — read all sixteen, wait, eight MFMAs per group was 650 clocks per group, as long as
            // the group's 16 KB take through the CU's fetch path; with 12 KB per group (8-column form) the chain was what bounded the phase
            v8s_s fa[2][kSGrp], fb[2][kSGrp];
            auto read_half = [&](int gi, int half) {
                const uint16_t* grp = smem + (size_t)(gi % kRing) * (kSGrp * kSSlot) + half * 2 * kSSlot;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        fa[half][qq * 2 + jj] = *reinterpret_cast<const v8s_s*>(grp + qq * kSSlot + lds_off(c, jj * 4 + g));
                        fb[half][qq * 2 + jj] = *reinterpret_cast<const v8s_s*>(grp + qq * kSSlot + lds_off(16 + c % CT, jj * 4 + g));
                    }
            };
            auto mfma_half = [&](int half) {
#pragma unroll
                for (int s4 = 0; s4 < kSGrp; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[half][s4], fb[half][s4], acc, 0, 0, 0);
            };
            if (wv == 0) {  // (a requester itself: wave 0 takes chunk parity 0, row octet 0)
#pragma unroll
                for (int gi = 0; gi < kSGroups; ++gi) {
                    wait_vmcnt(2 * (kSGroups - 1 - gi < kRing - 2 ? kSGroups - 1 - gi : kRing - 2));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // its reads of group gi - 1 are in registers
                    __builtin_amdgcn_s_barrier();
                    if (gi + kRing - 1 < kSGroups) request_group(gi + kRing - 1);
                    read_half(gi, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (gi > 0) mfma_half(1);  // the second half of group gi - 1
                    __builtin_amdgcn_sched_barrier(0);
                    read_half(gi, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_half(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma_half(1);
            } else {
#pragma unroll
                for (int gi = 0; gi < kSGroups; ++gi) {
                    wait_vmcnt(2 * (kSGroups - 1 - gi < kRing - 2 ? kSGroups - 1 - gi : kRing - 2));    // this wave's share of group gi has landed
                    __builtin_amdgcn_s_barrier();                                                        // ... everybody's; wave 0 is done with group gi - 1's slots
                    if (gi + kRing - 1 < kSGroups) request_group(gi + kRing - 1);                        // into the slots of group gi - 1
                }
            }
#endif
        } else {  // any other D (a multiple of 64): one chunk (4 KB = a 16-byte piece for each of 256 threads) per barrier pair
            const int cw = tid >> 8, lr = (tid & 255) >> 3, lp = tid & 7;
            const bool ld = cw == 0 && lr < kRS;
            const uint16_t* src = lr < 16 ? a.h + (size_t)min(rb * 16 + lr, R - 1) * D + lp * 8 : a.w + (size_t)(kA0 + j * CT + (lr - 16) % CT) * D + lp * 8;
            const int my_off = lds_off(lr, lp);
            for (int ch = 0; ch < nchunks; ++ch) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ld) v = *reinterpret_cast<const uint4*>(src + kchunk(ch) * kSK);
                __syncthreads();
                if (ld) *reinterpret_cast<uint4*>(smem + my_off) = v;
                __syncthreads();
                if (wv == 0) {
#pragma unroll
                    for (int jj = 0; jj < kSK / 32; ++jj) {
                        const v8s_s af1 = *reinterpret_cast<const v8s_s*>(smem + lds_off(c, jj * 4 + g));
                        const v8s_s bf1 = *reinterpret_cast<const v8s_s*>(smem + lds_off(16 + c % CT, jj * 4 + g));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bf1, acc, 0, 0, 0);
                    }
                }
            }
        }
        K3S_STAMP(1);
        if (wv == 0) {  // C/D layout: column = lane & 15, row = 4 (lane >> 4) + r; two columns + the tag per 64-bit agent-scope store
            // rows r, r + 1 packed; the neighbouring column's pair by a DPP quad permute (four ds_bpermute round trips sat on the critical path here)
            const unsigned z01 = pack_bf16x2(acc[0], acc[1]), z23 = pack_bf16x2(acc[2], acc[3]);
            const unsigned n01 = (unsigned)xor_lane_i<1>((int)z01), n23 = (unsigned)xor_lane_i<1>((int)z23);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned z = (((r & 2) ? z23 : z01) >> ((r & 1) * 16)) & 0xffffu;
                const unsigned zn = (((r & 2) ? n23 : n01) >> ((r & 1) * 16)) & 0xffffu;
                if (!(c & 1) && c < CT)
                    __hip_atomic_store(a.zs + (size_t)(rb * 16 + g * 4 + r) * (kNA / 2) + ((j * CT + c) >> 1), ((unsigned long long)a.tag << 32) | z | (zn << 16),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        K3S_STAMP(2);
        if (a.phases == 1) return;
    }
    // forward only: just the workgroups that leave statistics (j = 0 for a later fold) or fold them (workgroup 0) have something left to do
    if (!(a.dh != nullptr || a.gs != nullptr || blockIdx.x == 0 || (!a.publish && j == 0))) return;

    // ---- phase 2: statistics of the rows this workgroup needs (polled out of the scratch), then the gradient tile of its group's 16 NRB rows ----
    const bool pub_wg = a.publish && blockIdx.x == 0;
    const bool all_rows = a.ra.mode == VAA_LOSS_UPA || pub_wg;
    const bool writes_stats = a.publish ? pub_wg : (j == 0);  // who leaves the SliceStats (+ neutral parts) in K3's workspace, and for which rows
    const int wlo = a.publish ? 0 : rb * 16, whi = a.publish ? R : min(R, rb * 16 + 16);
    // a half wave's rows: 16 it + own_lr for row blocks it; the group's OWN blocks (the gradient's) come FIRST
    const int nblk = (R + 15) / 16;
    const int nit = all_rows ? max(nblk, rg * NRB + NRB) : NRB;
    auto it_of = [&](int i) { return i < NRB ? rg * NRB + i : (i - NRB < rg * NRB ? i - NRB : i); };  // every block of [0, nit) once: the own ones, then the others
    auto row_of = [&](int it) { return min(16 * it + own_lr, R - 1); };  // a row beyond the end repeats the last one (same values, same place)
    auto request = [&](int i, unsigned long long (&wq)[4]) {
        const unsigned long long* p = a.zs + (size_t)row_of(it_of(min(i, nit - 1))) * (kNA / 2) + hl * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto valid = [&](const unsigned long long (&wq)[4]) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) ok = ok && (unsigned)(wq[q] >> 32) == a.tag;
        return ok;
    };
    auto leave = [&](int it, float alse, float E, int pred) {  // one row's statistics: LDS for this workgroup, K3's workspace for the fold
        const int rr = row_of(it);
        if (hl == 0) {
            sAlse[rr] = alse; sE[rr] = E;
            if (writes_stats && rr >= wlo && rr < whi) {
                SliceStat ss;
                ss.alse = alse; ss.E = E; ss.pred = pred; ss.pad = 0;
                a.ra.slice[rr] = ss;
                PartStat nz;  // neutral element of the fold: "no full-vocabulary statistics on this step" (rows_fold)
                nz.m = -INFINITY; nz.s = 0.0f; nz.zlab = -INFINITY; nz.amax = 0x7fffffff;
                for (int q = 0; q < a.ra.split; ++q) a.ra.part[(size_t)rr * a.ra.split + q] = nz;
            }
        }
    };
    auto give_up = [&]() {  // NaN statistics and gradient (never stale ones) AND the process-wide failure word (vaa_async_error)
        if (lane == 0) {
            bar_ok = 0;
            if (a.err_word) __hip_atomic_store(a.err_word, VAA_ASYNC_K3_HANDOVER_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    auto grad_row = [&](int u, const float (&x)[8], float alse, float E, float kE) {  // g = kE p_a ((a + 1) - E), bf16, into the MFMA operand tile (+ the test output)
        const int row = (rg * NRB + u) * 16 + own_lr;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = row < Rn ? kE * expf(x[e] - alse) * ((float)(hl * 8 + e + 1) - E) : 0.0f;
        uint4 pk;
        pk.x = pack_bf16x2(o[0], o[1]);
        pk.y = pack_bf16x2(o[2], o[3]);
        pk.z = pack_bf16x2(o[4], o[5]);
        pk.w = pack_bf16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(gt + (u * 16 + own_lr) * kSGS + hl * 8) = pk;
        if (a.gs && j == 0 && u == mb && row < R) *reinterpret_cast<uint4*>(a.gs + (size_t)row * kNA + hl * 8) = pk;
    };
    if (tid == 0) bar_ok = 1;
    // wave 0 is past its fragment reads of phase 1 (it left the loop through its last barrier-free group): before anybody writes the tiles that
    // reuse the ring, one barrier — raw, behind an LDS-only wait
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    K3S_STAMP(3);
    // ---- phase 3's B operand: fragments of the transposed slice (128 KB / NRB per workgroup), requested in QUARTERS between the stages of the
    //      statistics below. All sixteen instructions per wave at once (128 KB per CU) stall the wave's issue behind the CU's fetch rate: behind
    //      the poll they held the statistics back by 2 us, in front of it the poll's words came back 2.9 us late. The workgroup's 16 output tiles:
    //      NRB row tiles x kDW / 16 column tiles; a wave's two tiles share their column tile when NRB > 1, so it fetches ONE tile's fragments ----
    constexpr int kDT = kDW / 16;                   // column tiles
    constexpr int kTPW = NRB * kDT / (kST / 64);    // output tiles per wave: 2 (one in the 8-column form)
    constexpr int kNB = NRB == 1 ? kTPW : 1;        // distinct column tiles among them
    static_assert(kTPW == 1 || kTPW == 2, "");
    const int dbase = (mb * kJ + j) * kDW;
    const int dt0 = NRB == 1 ? kTPW * wv : wv % kDT, rt0 = NRB == 1 ? 0 : (wv / kDT) * 2;  // tiles (rt0, dt0), (rt0 + 1, dt0) — or (0, dt0) [, (0, dt0 + 1)] at NRB = 1
    v8s_s bfr[kNB][kNA / 32];
    const uint16_t* bsrc[kNB];
#pragma unroll
    for (int t = 0; t < kNB; ++t) bsrc[t] = a.wt + (size_t)min(dbase + (dt0 + t) * 16 + c, D - 1) * kNA + g * 8;  // columns beyond D re-read the last one: never stored
    auto issue_b = [&](auto qc) {
        constexpr int kPerQ = kNB * (kNA / 32) / 4, q0 = decltype(qc)::value * kPerQ;
        if (a.dh) {
#pragma unroll
            for (int i = q0; i < q0 + kPerQ; ++i) bfr[i / (kNA / 32)][i % (kNA / 32)] = *reinterpret_cast<const v8s_s*>(bsrc[i / (kNA / 32)] + (i % (kNA / 32)) * 32);
        }
    };
    unsigned long long w0[NRB][4];
    // (the first poll in front of the barrier above — its round trip under the wait for wave 0 — is WORSE: 12.5 against 9.6 us at 128 rows; seven waves per
    //  workgroup poll early, miss, and their uncached re-reads delay the producers' stores;
    //  a pause of 0.2-0.3 us in front of it instead — so that it does not come back empty — : 10.1 against 10.9 us behind a 1 GiB copy, nothing warm,
    //  12.6 against 12.8 us in the bs=64 step (A B B A, within the noise): not kept)
#pragma unroll
    for (int u = 0; u < NRB; ++u) request(u, w0[u]);
    bool own_gave_up = a.max_polls < 0;  // (test hook VAA_K3_HANDOVER_POLLS=-1: the failure path, deterministically)
    if (!own_gave_up) {  // poll with NOTHING else in flight: loads return in order, a retry must not queue behind other requests
        int polls = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < NRB; ++u) ok = ok && valid(w0[u]);
            if (__all(ok)) break;
            if (++polls > a.max_polls) { own_gave_up = true; break; }  // a launch that never became fully resident gives up instead of hanging
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int u = 0; u < NRB; ++u) request(u, w0[u]);
        }
    }
    K3S_STAMP(4);
    issue_b(std::integral_constant<int, 0>{});
    constexpr int kPair = CT == 8 ? 1 : 2;  // row blocks of the other rows reduced side by side (one in the 8-column form: registers for two workgroups per CU)
    unsigned long long wpre[kPair][4];
#pragma unroll
    for (int u = 0; u < kPair; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) wpre[u][q] = 0ull;
    if (all_rows && NRB < nit) {  // UPA / the publishing workgroup: the next blocks' words travel while the own rows are reduced
#pragma unroll
        for (int u = 0; u < kPair; ++u) request(NRB + u, wpre[u]);
    }
    float xo[NRB][8], own_alse[NRB], own_E[NRB];
    {   // the own rows, their dependent shuffle chains side by side
        if (own_gave_up) give_up();
        int pred[NRB];
#pragma unroll
        for (int u = 0; u < NRB; ++u) decode_logits8(w0[u], xo[u]);
        slice_stats_half<NRB>(xo, hl, own_alse, own_E, pred, issue_b);  // (quarters 1, 2, 3 of the B fragments go out between its stages)
#pragma unroll
        for (int u = 0; u < NRB; ++u) {
            if (own_gave_up) { own_alse[u] = __uint_as_float(0x7fc00000u); own_E[u] = own_alse[u]; }
            leave(rg * NRB + u, own_alse[u], own_E[u], pred[u]);
            if (a.ra.mode == VAA_LOSS_UADA_DDP && (a.dh || a.gs)) {  // the gradient needs this row and the row COUNT only: straight from the registers
                float kE = 0.0f;
                if (me[u].lab > 2 && nact > 0) {
                    const double q = (double)own_E[u] / 256.0, t = (me[u].lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10)
                    kE = (float)((double)a.ra.w * a.ra.w * 2.0 * (q - t) / nact / 256.0);
                }
                if (own_gave_up) kE = __uint_as_float(0x7fc00000u);  // NaN gradient, never a stale one
                grad_row(u, xo[u], own_alse[u], own_E[u], kE);
            }
        }
    }
    K3S_STAMP(5);
    if (all_rows) {
        for (int i0 = NRB; i0 < nit; i0 += kPair) {  // the other rows (UPA's batch means, the publishing workgroup's fold): kPair blocks side by side
            unsigned long long wq[kPair][4];
            if (i0 == NRB) {  // the first ones were requested in front of the own rows' statistics (below the own poll): their round trip is over
#pragma unroll
                for (int u = 0; u < kPair; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) wq[u][q] = wpre[u][q];
            } else {
#pragma unroll
                for (int u = 0; u < kPair; ++u) request(i0 + u, wq[u]);
            }
            int polls = 0;
            bool gave_up = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < kPair; ++u) ok = ok && valid(wq[u]);
                if (__all(ok)) break;
                if (++polls > a.max_polls) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int u = 0; u < kPair; ++u) request(i0 + u, wq[u]);
            }
            if (gave_up) give_up();
            float x[kPair][8], alse[kPair], E[kPair];
            int pred[kPair];
#pragma unroll
            for (int u = 0; u < kPair; ++u) decode_logits8(wq[u], x[u]);
            slice_stats_half<kPair>(x, hl, alse, E, pred);
#pragma unroll
            for (int u = 0; u < kPair; ++u) {
                if (i0 + u >= nit) continue;
                if (gave_up) { alse[u] = __uint_as_float(0x7fc00000u); E[u] = alse[u]; }
                leave(it_of(i0 + u), alse[u], E[u], pred[u]);
            }
        }
    }
    if (a.ra.mode == VAA_LOSS_UPA) {  // UPA.py:375-387: the gradient needs the batch mean of ||e' - l'||, i.e. every row's E
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool poisoned = !bar_ok;
        // the fold in rows_fold's order (thread t takes row t; lanes by xor-shuffle, then waves in order): rows_fold<256> has thread t of 256 take
        // rows t, t + 256, ...; R' <= 128 leaves the upper waves empty, so the first four waves' sums in wave order are its bits (adding the
        // other waves' exact zeros changes nothing)
        double accd[2] = {0, 0};  // rows_fold's acc[5], acc[6]: each sum is reduced on its own there, so two of its seven give the same bits
        for (int rr = tid; rr < Rn; rr += kST) {
            const RowMap m = rm[rr];
            if (m.ord == 0 && rr + 2 < Rn) {
                Upa3 u3;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    RowStat t;
                    t.E = sE[rr + q];
                    t.lab = rm[rr + q].lab;
                    u3.set(q, t);
                }
                double c1, nd;
                u3.terms(c1, nd);
                accd[0] += c1;
                accd[1] += nd;
            }
        }
        block_sums<2, kST>(accd, reinterpret_cast<double (*)[2]>(&shf[0][0]));
        const double aux1 = 1.0 / (accd[1] / a.ra.B + 1e-3);  // UPA.py:384
        if (a.dh || a.gs) {
#pragma unroll
            for (int u = 0; u < NRB; ++u) {
                const int row = (rg * NRB + u) * 16 + own_lr;
                float kE = 0.0f;
                if (row < Rn && me[u].ord < 3 && row - me[u].ord >= 0 && row - me[u].ord + 2 < Rn) {
                    Upa3 u3;
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        RowStat t;
                        t.E = sE[row - me[u].ord + q];
                        t.lab = rm[row - me[u].ord + q].lab;
                        u3.set(q, t);
                    }
                    kE = (float)(u3.dE(me[u].ord, (double)a.ra.alpha, (double)a.ra.beta, aux1, a.ra.B) / 255.0);
                }
                if (poisoned) kE = __uint_as_float(0x7fc00000u);
                grad_row(u, xo[u], own_alse[u], own_E[u], kE);
            }
        }
    }
    K3S_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    K3S_STAMP(7);

    // ---- phase 3: dH tile = g [16 NRB, 256] x slice [256, these kDW columns of D] ----
    if (a.dh) {
        v4f_s acc3[kTPW];
#pragma unroll
        for (int t = 0; t < kTPW; ++t) acc3[t] = (v4f_s){0.f, 0.f, 0.f, 0.f};
        const int rt1 = NRB == 1 ? 0 : rt0 + 1;
#pragma unroll
        for (int ks = 0; ks < kNA / 32; ++ks) {
            const v8s_s af0 = *reinterpret_cast<const v8s_s*>(gt + (rt0 * 16 + c) * kSGS + ks * 32 + g * 8);
            acc3[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr[0][ks], acc3[0], 0, 0, 0);
            if (kTPW == 2) {
                const v8s_s af1 = NRB == 1 ? af0 : *reinterpret_cast<const v8s_s*>(gt + (rt1 * 16 + c) * kSGS + ks * 32 + g * 8);
                acc3[kTPW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr[kNB - 1][ks], acc3[kTPW - 1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < kTPW; ++t) {
            const int rt = NRB == 1 ? 0 : rt0 + t, dt = NRB == 1 ? dt0 + t : dt0;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const unsigned pr = pack_bf16x2(acc3[t][r], acc3[t][r + 1]);
                ot[(rt * 16 + g * 4 + r) * kOS + dt * 16 + c] = (uint16_t)(pr & 0xffffu);
                ot[(rt * 16 + g * 4 + r + 1) * kOS + dt * 16 + c] = (uint16_t)(pr >> 16);
            }
        }
        K3S_STAMP(8);
        __syncthreads();
        // 16 NRB rows x kDW columns = 4096 elements (2048 in the 8-column form) = one 16-byte piece per thread
        const int orow = tid / (kDW / 8), col = (tid % (kDW / 8)) * 8, d = dbase + col, grow = rg * NRB * 16 + orow;
        if (orow < 16 * NRB && grow < R && d < D) *reinterpret_cast<uint4*>(a.dh + (size_t)grow * D + d) = *reinterpret_cast<const uint4*>(ot + orow * kOS + col);
    }
    K3S_STAMP(9);
    // ---- publication (callers without a step epilogue): workgroup 0 folds the rows it has just written into scalars[8] + the prediction maps ----
    if (pub_wg) {
        __syncthreads();  // this workgroup's SliceStat / PartStat stores are visible to all of its threads
        (void)rows_fold<kST>(a.ra, true, shf);  // (512 threads: the upper waves add exact zeros to rows_fold<256>'s sums — the same bits)
    }
}

// wt[d][a] = w[31744 + a][d]: one 64 x 64 tile per workgroup through LDS
__global__ __launch_bounds__(256) void head_slice_pack_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wt, int D) {
    __shared__ uint16_t tile[64][66];
    const int d0 = blockIdx.x * 64, a0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int ar = e >> 6, dc = e & 63;
        tile[ar][dc] = (d0 + dc < D) ? w[(size_t)(kA0 + a0 + ar) * D + d0 + dc] : (uint16_t)0;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
        const int dr = e >> 6, ac = e & 63;
        if (d0 + dr < D) wt[(size_t)(d0 + dr) * kNA + a0 + ac] = tile[ac][dr];
    }
}

// the ring (the gradient / output tiles reuse it): 128 KB, or 72 KB in the 8-column form
static size_t slice_lds_bytes(int ct) { return (size_t)(ct == 16 ? kSRing : kSRing8) * kSGrp * (16 + ct) * kSK * sizeof(uint16_t); }

// instantiations: row-block groups of 1 / 2 / 4 in the 16-column form, and the 8-column form
static int slice_inst(int nrb_group, int ct) { return ct == 8 ? 3 : (nrb_group == 1 ? 0 : (nrb_group == 2 ? 1 : 2)); }
static const void* slice_kernel_fn(int inst) {
    switch (inst) {
        case 0: return (const void*)head_slice_kernel<1, 16>;
        case 1: return (const void*)head_slice_kernel<2, 16>;
        case 2: return (const void*)head_slice_kernel<4, 16>;
        default: return (const void*)head_slice_kernel<1, 8>;
    }
}

// resident workgroups of head_slice_kernel<NRB> on the current device (occupancy x CUs), queried once per device and instantiation; 0 = unknown
static long slice_resident_slots(int inst) {
    static std::atomic<long> slots[4][16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return 0; }
    long have = slots[inst][dev].load(std::memory_order_relaxed);
    if (have == 0) {
        int cus = 0, per_cu = 0;
        hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) {
            if (inst == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_slice_kernel<1, 16>, kST, slice_lds_bytes(16));
            else if (inst == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_slice_kernel<2, 16>, kST, slice_lds_bytes(16));
            else if (inst == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_slice_kernel<4, 16>, kST, slice_lds_bytes(16));
            else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_slice_kernel<1, 8>, kST, slice_lds_bytes(8));
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            have = -1;
        } else {
            have = (long)cus * per_cu;
        }
        slots[inst][dev].store(have, std::memory_order_relaxed);
    }
    return have > 0 ? have : 0;
}

static void slice_launch(int inst, unsigned grid, size_t lds, hipStream_t st, const SliceHeadArgs& a) {
    if (inst == 0) launch_k("head_slice_kernel<1>", head_slice_kernel<1, 16>, dim3(grid), dim3(kST), lds, st, a);
    else if (inst == 1) launch_k("head_slice_kernel<2>", head_slice_kernel<2, 16>, dim3(grid), dim3(kST), lds, st, a);
    else if (inst == 2) launch_k("head_slice_kernel<4>", head_slice_kernel<4, 16>, dim3(grid), dim3(kST), lds, st, a);
    else launch_k("head_slice_kernel<1,8>", head_slice_kernel<1, 8>, dim3(grid), dim3(kST), lds, st, a);
}

}  // namespace vaa

extern "C" size_t vaa_loss_rows_ws_bytes(int R);

extern "C" int vaa_head_slice_applies(int R, int D, int V) {
    // V <= 32,768: the fold arithmetic this kernel shares with rows_finish_kernel is its 256-thread form
    return (R > 0 && R <= vaa::kSRowsMax && D >= vaa::kSK && (D % vaa::kSK) == 0 && D <= vaa::kSDMax && V >= vaa::kA0 + vaa::kNA && (V % 8) == 0 &&
            V <= 32768) ? 1 : 0;
}

namespace vaa {
// row blocks per group of phases 2 / 3 and the padded block count (a multiple of it): 1, 2 or 4 blocks (R' <= 16, <= 32, more)
static void slice_groups(int R, int& nrb_group, int& nblk_pad) {
    const int nblk = (R + 15) / 16;
    // ONE row block per workgroup: grouping 2 / 4 blocks (VAA_K3S_GROUP) divides phase 3's B fetch (128 KB per workgroup) by as much but multiplies
    // the redundant statistics + gradient work, which is VALU-bound (16 accurate expf per logit octet and wave): 13.1 -> 26 us at four blocks
    nrb_group = 1;
    const char* ev = getenv("VAA_K3S_GROUP");
    if (ev && (ev[0] == '2' || ev[0] == '4')) nrb_group = (ev[0] - '0') <= nblk || nblk >= 3 ? (ev[0] - '0') : nblk;
    if (nrb_group == 4 && nblk < 3) nrb_group = nblk >= 2 ? 2 : 1;
    nblk_pad = (nblk + nrb_group - 1) / nrb_group * nrb_group;
}
// action columns per workgroup in phase 1: 8 (32 workgroups per row block) unless VAA_K3S_COLS=16 or a row-block group asks for the 16-column form
static int slice_cols(int nrb_group) {
    const char* ev = getenv("VAA_K3S_COLS");
    return (nrb_group > 1 || (ev && ev[0] == '1' && ev[1] == '6')) ? 16 : 8;
}
}  // namespace vaa

extern "C" size_t vaa_head_slice_ws_bytes(int R) {
    if (R <= 0) return 0;
    int g = 1, nb = 1;
    vaa::slice_groups(R, g, nb);
    return (size_t)nb * 16 * (vaa::kNA / 2) * sizeof(unsigned long long);  // two logits + the launch tag per 64-bit word, padded to whole groups of row blocks
}

extern "C" int vaa_head_slice_pack(const uint16_t* w_head, int D, int V, uint16_t* w_slice_t, void* stream) {
    using namespace vaa;
    const char* who = "vaa_head_slice_pack";
    if (!w_head || !w_slice_t) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (D <= 0 || V < kA0 + kNA) {
        set_error("%s: bad sizes (D=%d V=%d)", who, D, V);
        return VAA_E_INVALID;
    }
    VAA_LAUNCH(head_slice_pack_kernel, dim3((unsigned)((D + 63) / 64), kNA / 64), dim3(256), 0, (hipStream_t)stream, w_head, w_slice_t, D);
    return check_launch(who);
}

extern "C" int vaa_head_slice_fwd_bwd(const uint16_t* hidden, const uint16_t* w_head, const uint16_t* w_slice_t, int D, const void* rowmap, int R, int B,
                                      int L, int V, int mode, const float* params, uint16_t* dhidden, uint16_t* grad_slice, void* loss_ws,
                                      size_t loss_ws_bytes, float* scalars, int32_t* pred_tokens, int32_t* pred_full_tokens, void* ws, size_t ws_bytes,
                                      void* stream) {
    using namespace vaa;
    const char* who = "vaa_head_slice_fwd_bwd";
    if (!hidden || !w_head || !rowmap || !params || !loss_ws || !ws || (dhidden && !w_slice_t)) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (mode != VAA_LOSS_UADA_DDP && mode != VAA_LOSS_UPA) {
        set_error("%s: mode %d has a cross-entropy term — its loss does not live in the action columns (vaa_head_loss_rows_stats / the LM-head GEMM)", who, mode);
        return VAA_E_UNSUPPORTED;
    }
    if (!vaa_head_slice_applies(R, D, V) || B <= 0 || L <= 1 || (long)R > (long)B * (L - 1)) {
        set_error("%s: shape not covered (R=%d <= %d rows, D=%d a multiple of %d up to %d, V=%d <= 32768; B=%d L=%d)", who, R, kSRowsMax, D, kSK,
                  kSDMax, V, B, L);
        return VAA_E_UNSUPPORTED;
    }
    if ((((uintptr_t)hidden) | ((uintptr_t)w_head) | ((uintptr_t)w_slice_t) | ((uintptr_t)dhidden) | ((uintptr_t)grad_slice) | ((uintptr_t)ws)) & 15u) {
        set_error("%s: hidden, w_head, w_slice_t, dhidden, grad_slice and ws must be 16-byte aligned", who);
        return VAA_E_INVALID;
    }
    if (loss_ws_bytes < vaa_loss_rows_ws_bytes(R) || ws_bytes < vaa_head_slice_ws_bytes(R)) {
        set_error("%s: workspace too small (loss %zu of %zu B, slice %zu of %zu B)", who, loss_ws_bytes, vaa_loss_rows_ws_bytes(R), ws_bytes,
                  vaa_head_slice_ws_bytes(R));
        return VAA_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    SliceHeadArgs a = {};
    a.h = hidden; a.w = w_head; a.wt = w_slice_t;
    a.zs = (unsigned long long*)ws; a.gs = grad_slice; a.dh = dhidden;
    a.ra.logits = nullptr; a.ra.rowmap = (const int*)rowmap;
    a.ra.part = (PartStat*)loss_ws; a.ra.slice = (SliceStat*)((char*)loss_ws + (size_t)R * 4 * sizeof(PartStat));
    a.ra.grad = nullptr; a.ra.scalars = scalars; a.ra.pred_tokens = pred_tokens; a.ra.pred_full = pred_full_tokens;
    a.ra.R = R; a.ra.B = B; a.ra.L = L; a.ra.V = V; a.ra.mode = mode; a.ra.split = rows_split(R, V); a.ra.grad_slice = 1;
    a.ra.ldz = kNA; a.ra.zcol0 = kA0;
    a.ra.w = params[0]; a.ra.alpha = params[1]; a.ra.beta = params[2]; a.ra.scale = params[3];
    a.D = D; a.publish = scalars ? 1 : 0;
    static_assert(16 * 4 * kSGS + 16 * 4 * (kNA / 4 + 8) <= kSRing * kSGrp * 32 * kSK && 16 * kSGS + 16 * (kSDMax / 32 + 8) <= kSRing8 * kSGrp * 24 * kSK,
                  "the tiles reuse the ring");
    int nrb_group = 1, nblk_pad = 1;
    slice_groups(R, nrb_group, nblk_pad);
    const int ct = slice_cols(nrb_group), inst = slice_inst(nrb_group, ct);
    const size_t lds = slice_lds_bytes(ct);
    static std::atomic<unsigned long long> attr_done[4];  // [instantiation] bit = device ordinal: the dynamic-LDS opt-in (128 / 72 KB) is set once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (lds > 64 * 1024 && !(attr_done[inst].load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(slice_kernel_fn(inst), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute failed", who);
            return VAA_E_LAUNCH;
        }
        attr_done[inst].fetch_or(bit, std::memory_order_acq_rel);
    }
    const unsigned grid = (unsigned)nblk_pad * (unsigned)(kNA / ct);
    // the launch's tag: unique in the process (one counter for all streams and devices), scrambled so that no plausible stale content of the
    // scratch (small integers, bf16 pairs, an old launch's tag) equals it
    static std::atomic<unsigned> tag_counter{0u};
    a.tag = (tag_counter.fetch_add(1u, std::memory_order_relaxed) + 1u) * 2654435761u | 0x80000001u;
    a.err_word = async_error_word();
    a.max_polls = 1 << 22;
    const char* pv = getenv("VAA_K3_HANDOVER_POLLS");  // test hook: 0 makes every waiting workgroup give up at once
    if (pv && *pv) a.max_polls = atoi(pv);
    // ONE launch when a grid that waits on itself is admissible: the device keeps at least TWICE the grid resident (a second waiting grid of
    // another process still finds room), the stream is not being captured and no other stream of this process has a waiting grid in flight;
    // else the same kernel twice (phase 1, then phases 2 + 3): the same bits. VAA_K3S_ONE_LAUNCH=0 forces two.
    const char* ev = getenv("VAA_K3S_ONE_LAUNCH");
    bool one = !(ev && ev[0] == '0');
    if (one) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const hipError_t ce = hipStreamIsCapturing(st, &cs);
        if (ce != hipSuccess) (void)hipGetLastError();
        const bool capturing = ce != hipSuccess || cs != hipStreamCaptureStatusNone;
        one = !capturing && 2l * grid <= slice_resident_slots(inst) && rows_one_pass_stream_ok(st);
    }
    const char* dbg = getenv("VAA_K3S_DEBUG_PHASES");  // measurement hook (tools/k3s_bench.py): run phase 1 or phases 2 + 3 alone
    if (dbg && (dbg[0] == '1' || dbg[0] == '2')) {
        static unsigned last_tag;  // phase 2 alone polls for the words the last phase-1 run left
        if (dbg[0] == '2') a.tag = last_tag;
        last_tag = a.tag;
        a.phases = dbg[0] - '0';
        slice_launch(inst, grid, lds, st, a);
        return check_launch(who);
    }
    if (one) {
        a.phases = 3;
        slice_launch(inst, grid, lds, st, a);
        return check_launch(who);
    }
    a.phases = 1;
    slice_launch(inst, grid, lds, st, a);
    int rc = check_launch("vaa_head_slice_fwd_bwd(logits)");
    if (rc != VAA_OK) return rc;
    a.phases = 2;
    slice_launch(inst, grid, lds, st, a);
    return check_launch("vaa_head_slice_fwd_bwd(gradient)");
}
