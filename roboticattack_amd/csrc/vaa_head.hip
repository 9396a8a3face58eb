// vaa_head.hip — SURVEY.md section 8f-2 as the survey wrote it: the LM head on the labelled rows FUSED with K3's statistics.
//
// Replaces `logits = lm_head(hidden_rows)` (modeling_prismatic.py:404-415 -> HF Llama's lm_head, bf16) + the statistics pass of K3
// (vaa_loss_rows_stats: HF's CE terms and weighted_loss's soft-argmax, UADA_ddp.py:99-124) for the data-parallel UADA step, whose gradient
// lives in the 256 action columns: the [R',V] logits are never written to memory.
//
//   head_stats_kernel<NRB>  grid = ceil(V / 128) workgroups of 8 waves; workgroup w owns the 128 vocabulary columns [128 w, 128 w + 128):
//       * the kernel is a WEIGHT STREAM — 263 MB at V = 32,064, D = 4,096, read from HBM exactly once; its floor is 263 MB / HBM bandwidth.
//         Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) in FULL 128-byte
//         lines: a DMA instruction moves 8 rows x 128 B of a 64-wide k-chunk (64-byte fragment-shaped loads straight into MFMA operands
//         streamed at 3.0-4.5 TB/s whatever the prefetch depth, full lines reach 6.2 TB/s: profiles/r04_head_fused.txt). A ring of four
//         LDS slots of [hidden rows | 128 weight rows]: chunk c + 3 is requested in step c into the slot chunk c - 1 was read from; counted
//         vmcnt waits + ONE raw s_barrier per chunk (a __syncthreads would drain the requests in flight); weights with the nontemporal
//         policy. Workgroup w walks K starting at chunk 5 w mod 64: all workgroups reading the same k offset of weight rows 8 KB apart at
//         once put the chip's requests on a few memory channels (58 -> 43 us at R' = 16);
//       * the hidden rows H [R' <= 128, D] (1 MB, L2-resident) are read by every workgroup the same way; mfma_f32_16x16x32_bf16, a wave
//         multiplies all rows against its 16 columns (R' <= 64) or half the rows against 32 columns (R' = 128: 6 fragment reads per 8 MFMAs);
//       * epilogue: the accumulators are rounded to bf16 (what the reference's bf16 head hands to `.float()`), laid out as a [R', 128] tile
//         in LDS and reduced per row to {max, sum exp, argmax, label logit} = one PartStat per (row, workgroup); the two workgroups that
//         own the action columns 31744..31999 also leave those logits in a [R', 256] fp32 buffer.
//       Above 64 rows the kernel is bound by the LINE RATE of a CU's memory path, not by HBM: every workgroup pulls all of H next to its
//       share of W (256 lines of 128 B per chunk at R' = 128 against 160 at R' <= 32) and a CU sustains ~300 lines/us in this mix —
//       42 / 44 / 49 / 55 us back to back at 32 / 64 / 96 / 128 rows whatever the staging form or the prefetch depth (profiles/r04_head_fused.txt).
//   head_finish_kernel      grid = R' workgroups: folds a row's ceil(V / 128) PartStats into ONE (stored in K3's workspace layout, the other
//       parts neutral), computes the action-slice statistics with the arithmetic of rows_stats_kernel (same bits for the same logits) and
//       — UADA_DDP — writes the gradient slice. vaa_step_epilogue then folds the rows exactly as it does behind vaa_loss_rows_stats.
#include <atomic>

#include "vaa_common.h"
#include "vaa_rows.h"

namespace vaa {

typedef short v8s_h __attribute__((ext_vector_type(8)));
typedef float v4f_h __attribute__((ext_vector_type(4)));

#ifndef VAA_HEAD_WAVES
#define VAA_HEAD_WAVES 8
#endif
constexpr int kHT = VAA_HEAD_WAVES * 64;  // threads per workgroup: 8 waves (two per SIMD: one wave's fragment reads hide behind the other's MFMAs)
constexpr int kHCols = kHeadCols;    // vocabulary columns per workgroup (128)
constexpr int kHK = 64;              // k-chunk: one 128-byte line of every weight row and of every hidden row
constexpr int kHSA = kHK;            // LDS row (bf16 elements): 128 B, UNPADDED — the eight 16-byte pieces of row r sit at slot (piece ^ (r & 7)).
                                     // The hardware serves a ds_read_b128 in groups of 16 lanes that mix two k-groups ({0-3, 12-15} of one with {4-11} of
                                     // the next): with rows padded to 144 B 7 of 16 lanes of every group met a bank another lane held (SQ_LDS_BANK_CONFLICT
                                     // = 36 % of SQ_LDS_IDX_ACTIVE); with the XOR placement the 16 lanes of a group cover the 16 slots of 256 B exactly once
constexpr int kHTileS = kHCols + 4;  // padded row of the fp32 logits tile
constexpr int kHRowsMax = 128;
constexpr int kHRing = 4;            // LDS slots: chunks ch + 1, ch + 2 (and, once requested, ch + 3) are in flight while chunk ch is multiplied
struct HeadArgs {
    const uint16_t* h;      // [R, D] bf16 hidden rows (final norm applied)
    const uint16_t* w;      // [V, D] bf16 LM-head weight
    const int* rowmap;      // K3's row map {R, #action rows, 0, 0} + RowMap[R]
    PartStat* part;         // [R][nwg]
    uint16_t* slice_logits; // [R][256] action-column logits, bf16
    uint16_t* logits_dbg;   // [R][V] bf16 or nullptr (tests)
    int R, D, V, nwg;
};

template <int NRB>
__global__ __launch_bounds__(kHT) void head_stats_kernel(HeadArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the kernel's stub (it does not know the LDS-DMA builtin)
    extern __shared__ __align__(16) unsigned char head_smem[];
    constexpr int ROWS = NRB * 32, NQ = NRB * 2, NW = VAA_HEAD_WAVES;
    constexpr int kHCB = kHCols / 16 / NW;  // 16-column blocks per wave
    // waves = (column group, row half): with 128 rows a wave takes HALF the rows against TWO column blocks — 6 fragment reads per 8 MFMAs instead of 9
    constexpr int RH = (NRB >= 4 && kHCB == 1) ? 2 : 1, CBW = kHCB * RH, NQW = NQ / RH;
    constexpr int RING = kHRing, TR = ROWS + kHCols;          // ring slots; rows per slot: [hidden rows | weight rows]
    uint16_t* ring = reinterpret_cast<uint16_t*>(head_smem);  // [RING][TR][kHSA]
    float* tile = reinterpret_cast<float*>(head_smem);        // epilogue: [ROWS][kHTileS]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * kHCols;
    const int nchunks = a.D / kHK;
    // k-chunk order: workgroup w starts at chunk (5 w) mod nchunks and wraps around (all workgroups reading the same k offset of weight rows 8 KB
    // apart at the same time puts the requests of the whole chip on a few memory channels)
    const int kstart = (int)((blockIdx.x * 5u) % (unsigned)nchunks);
    auto kchunk = [&](int ch) { const int cc = min(ch, nchunks - 1) + kstart; return cc >= nchunks ? cc - nchunks : cc; };
    auto lds_off = [](int row, int piece) { return row * kHSA + ((piece ^ (row & 7)) << 3); };  // element offset of (row, 16-byte piece) in a slot
    const int cg = wv / RH, rh = wv % RH;
    v4f_h acc[CBW][NQW];
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int q = 0; q < NQW; ++q) acc[cb][q] = (v4f_h){0.f, 0.f, 0.f, 0.f};
    const int arow = rh * (ROWS / RH) + c, brow = ROWS + cg * (16 * CBW) + c;  // row blocks start at multiples of 16: (row & 7) == (c & 7) for every block

    // ---- LDS-DMA: instruction = 64 lanes x 16 B to (wave-uniform base) + 16 lane = 8 rows x 128 B of the UNPADDED slot image; lane l = (row l >> 3,
    //      slot l & 7) fetches piece (l & 7) ^ (row & 7): the XOR placement on the SOURCE side ----
    constexpr int NIH = ROWS / 8, IWH = (NIH + NW - 1) / NW, IWW = kHCols / 8 / NW, IW = IWH + IWW;  // instructions per wave and chunk: hidden + weights
    static_assert(kHCols / 8 % NW == 0, "weight rows split evenly over the waves");
    const int wvu = __builtin_amdgcn_readfirstlane(wv);  // the DMA destination (M0) is per wave
    const int lrow = lane >> 3, lpiece = (lane & 7) ^ lrow;
    const uint16_t* src[IW];
    int dsto[IW];  // element offset of the instruction's 8 rows inside a slot
#pragma unroll
    for (int it = 0; it < IWH; ++it) {
        const int i = min(wvu + it * NW, NIH - 1);  // fewer instructions than waves: the spare waves repeat the last one (same bytes, same place)
        src[it] = a.h + (size_t)min(i * 8 + lrow, a.R - 1) * a.D + lpiece * 8;  // rows beyond R' repeat the last row: never reduced
        dsto[it] = i * 8 * kHSA;
    }
#pragma unroll
    for (int it = 0; it < IWW; ++it) {
        const int i = wvu + it * NW;
        src[IWH + it] = a.w + (size_t)min(n0 + i * 8 + lrow, a.V - 1) * a.D + lpiece * 8;  // columns beyond V re-read the last row; never used
        dsto[IWH + it] = (ROWS + i * 8) * kHSA;
    }
    auto request = [&](int ch) {  // the request counts are static: behind the end the last chunk is requested again (never read)
        const int koff = kchunk(ch) * kHK;
        uint16_t* slot = ring + (size_t)(ch & (RING - 1)) * TR * kHSA;
#pragma unroll
        for (int it = 0; it < IW; ++it) {
            auto* dst = (__attribute__((address_space(3))) void*)(slot + dsto[it]);
            if (it < IWH) __builtin_amdgcn_global_load_lds(src[it] + koff, dst, 16, 0, 0);
            else __builtin_amdgcn_global_load_lds(src[it] + koff, dst, 16, 0, 2);  // weights are streamed once: nontemporal
        }
    };
#pragma unroll
    for (int ch = 0; ch < RING - 1; ++ch) request(ch);
    for (int ch = 0; ch < nchunks; ++ch) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * IW) : "memory");  // this wave's share of chunk ch has landed (requests complete in order)
        __builtin_amdgcn_s_barrier();                                           // ... and everybody else's; everybody is done reading chunk ch - 1
        request(ch + RING - 1);                                                 // into the slot of chunk ch - 1
        const uint16_t* sl = ring + (size_t)(ch & (RING - 1)) * TR * kHSA;
        // fragments of k-step jj + 1 are read before the MFMAs of k-step jj (two register sets)
        v8s_h af[2][NQW], bf[2][CBW];
        auto read_frags = [&](int set, int jj) {
#pragma unroll
            for (int cb = 0; cb < CBW; ++cb) bf[set][cb] = *reinterpret_cast<const v8s_h*>(sl + lds_off(brow + cb * 16, jj * 4 + g));
#pragma unroll
            for (int q = 0; q < NQW; ++q) af[set][q] = *reinterpret_cast<const v8s_h*>(sl + lds_off(arow + q * 16, jj * 4 + g));
        };
        read_frags(0, 0);
#pragma unroll
        for (int jj = 0; jj < kHK / 32; ++jj) {
            if (jj + 1 < kHK / 32) read_frags((jj + 1) & 1, jj + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQW; ++q)
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb) acc[cb][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jj & 1][q], bf[jj & 1][cb], acc[cb][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the spare requests behind the end must land before the LDS becomes the logits tile
    __syncthreads();

    // ---- epilogue: bf16-rounded logits -> LDS tile [ROWS][128]; C/D layout: column = lane & 15, row = 4 (lane >> 4) + r ----
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
        for (int q = 0; q < NQW; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(rh * (ROWS / RH) + q * 16 + g * 4 + r) * kHTileS + cg * (16 * CBW) + cb * 16 + c] = bf16_bits_to_f32(f32_to_bf16_bits(acc[cb][q][r]));
    __syncthreads();
    const int ncols = min(kHCols, a.V - n0);
    // per-row statistics: TPR (= 4 with 8 waves) threads per row, thread `part` takes the columns part, part + TPR, ...
    {
        constexpr int TPR = kHT / kHRowsMax;  // threads per row
        static_assert(kHT % kHRowsMax == 0 && TPR >= 1 && (TPR & (TPR - 1)) == 0 && TPR <= 64,
                      "VAA_HEAD_WAVES * 64 must be a power-of-two multiple of kHRowsMax: the per-row folds below are xor-shuffles over TPR lanes");
        const int row = tid / TPR, part = tid % TPR;
        const bool live = row < ROWS && row < a.R;
        float m = -INFINITY;
        int mi = 0x7fffffff;
        if (live) {
            const float* tr = tile + row * kHTileS;
            for (int cl = part; cl < ncols; cl += TPR) {  // increasing columns: the first maximum wins
                const float v = tr[cl];
                if (v > m) { m = v; mi = n0 + cl; }
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float s = 0.0f;
        if (live) {
            const float* tr = tile + row * kHTileS;
            for (int cl = part; cl < ncols; cl += TPR) s += expf(tr[cl] - m);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o, 64);
        if (live && part == 0) {
            PartStat ps;
            ps.m = m;
            ps.s = s;
            ps.amax = mi;
            ps.zlab = -INFINITY;
            if (row < a.rowmap[0]) {
                const int lab = reinterpret_cast<const RowMap*>(a.rowmap + 4)[row].lab;
                if (lab >= n0 && lab < n0 + ncols) ps.zlab = tile[row * kHTileS + (lab - n0)];
            }
            a.part[(size_t)row * a.nwg + blockIdx.x] = ps;
        }
    }
    const int rows_live = min(ROWS, a.R);
    if (n0 >= kA0 && n0 < kA0 + kNA) {  // the action columns (31744 is a multiple of 128: two whole workgroups)
        for (int idx = tid; idx < rows_live * kHCols; idx += kHT) {
            const int r = idx >> 7, cl = idx & 127;
            a.slice_logits[(size_t)r * kNA + (n0 - kA0) + cl] = (uint16_t)f32_to_bf16_bits(tile[r * kHTileS + cl]);  // (exact: the tile holds bf16-rounded values)
        }
    }
    if (a.logits_dbg) {
        for (int idx = tid; idx < rows_live * kHCols; idx += kHT) {
            const int r = idx >> 7, cl = idx & 127;
            if (cl < ncols) a.logits_dbg[(size_t)r * a.V + n0 + cl] = (uint16_t)f32_to_bf16_bits(tile[r * kHTileS + cl]);
        }
    }
#endif
}

struct HeadFinishArgs {
    const PartStat* part_in;    // [R][nwg]
    const uint16_t* slice_logits;  // [R][256] bf16
    const int* rowmap;
    PartStat* part_out;         // K3 workspace: [R][4], `split` parts per row are read by the fold
    SliceStat* slice_out;       // [R]
    uint16_t* grad_slice;       // [R][256] bf16 or nullptr (UADA_DDP)
    int R, nwg, split, mode;
    float w;
};

__global__ __launch_bounds__(256) void head_finish_kernel(HeadFinishArgs a) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ float redm[4], reds[4], redz[4];
    __shared__ int redi[4];
    // ---- fold the row's parts (workgroup order = column order: ties keep the lowest column) ----
    float m = -INFINITY, zl = -INFINITY, s_acc = 0.0f;
    int mi = 0x7fffffff;
    float pm[4], psum[4];  // up to 1024 parts (V <= 131,072)
    int np = 0;
    for (int t = tid; t < a.nwg; t += 256) {
        const PartStat p = a.part_in[(size_t)r * a.nwg + t];
        pm[np] = p.m; psum[np] = p.s; ++np;
        if (p.m > m || (p.m == m && p.amax < mi)) { m = p.m; mi = p.amax; }
        zl = fmaxf(zl, p.zlab);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        zl = fmaxf(zl, __shfl_xor(zl, o, 64));
    }
    if (lane == 0) { redm[wv] = m; redi[wv] = mi; redz[wv] = zl; }
    __syncthreads();
    float M = redm[0], Z = redz[0];
    int MI = redi[0];
    for (int q = 1; q < 4; ++q) {
        if (redm[q] > M || (redm[q] == M && redi[q] < MI)) { M = redm[q]; MI = redi[q]; }
        Z = fmaxf(Z, redz[q]);
    }
    for (int q = 0; q < np; ++q) s_acc += psum[q] * expf(pm[q] - M);
    s_acc = wave_sum(s_acc);
    if (lane == 0) reds[wv] = s_acc;
    __syncthreads();
    if (tid == 0) {
        PartStat o;
        o.m = M;
        o.s = (reds[0] + reds[1]) + (reds[2] + reds[3]);
        o.zlab = Z;
        o.amax = MI;
        a.part_out[(size_t)r * a.split] = o;
        PartStat nz;  // neutral element of the fold's combination
        nz.m = -INFINITY; nz.s = 0.0f; nz.zlab = -INFINITY; nz.amax = 0x7fffffff;
        for (int q = 1; q < a.split; ++q) a.part_out[(size_t)r * a.split + q] = nz;
    }
    if (wv != 0) return;
    // ---- action slice: the arithmetic of rows_stats_kernel's bf16 instantiation (32 lanes x 8 logits), so that the same logits give the same bits ----
    constexpr int N = 8, nthr = kNA / N;
    const bool own = lane < nthr;
    float x[N];
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = -INFINITY;
    if (own) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.slice_logits + (size_t)r * kNA + lane * N);
        x[0] = bf16_bits_to_f32(v.x & 0xffffu); x[1] = bf16_bits_to_f32(v.x >> 16); x[2] = bf16_bits_to_f32(v.y & 0xffffu); x[3] = bf16_bits_to_f32(v.y >> 16);
        x[4] = bf16_bits_to_f32(v.z & 0xffffu); x[5] = bf16_bits_to_f32(v.z >> 16); x[6] = bf16_bits_to_f32(v.w & 0xffffu); x[7] = bf16_bits_to_f32(v.w >> 16);
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
        a.slice_out[r] = ss;
    }
    if (a.mode == VAA_LOSS_UADA_DDP && a.grad_slice && own) {  // gradient of w^2 * mean((E / 256 - t)^2): this row and the row COUNT only
        RowMap me = {0, 0, -1, 0};
        if (r < a.rowmap[0]) me = reinterpret_cast<const RowMap*>(a.rowmap + 4)[r];
        const int nact = a.rowmap[1];
        float kE = 0.0f;
        if (me.lab > 2 && nact > 0) {
            const double rr = (double)E / 256.0, t = (me.lab > 31872) ? 0.0 : 1.0;  // UADA.py:390-394 (A-D10)
            kE = (float)((double)a.w * a.w * 2.0 * (rr - t) / nact / 256.0);
        }
        const float alse = am + logf(es);
        float o[N];
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] = kE * expf(x[e] - alse) * ((float)(lane * N + e + 1) - E);
        uint4 pk;
        pk.x = f32_to_bf16_bits(o[0]) | (f32_to_bf16_bits(o[1]) << 16);
        pk.y = f32_to_bf16_bits(o[2]) | (f32_to_bf16_bits(o[3]) << 16);
        pk.z = f32_to_bf16_bits(o[4]) | (f32_to_bf16_bits(o[5]) << 16);
        pk.w = f32_to_bf16_bits(o[6]) | (f32_to_bf16_bits(o[7]) << 16);
        *reinterpret_cast<uint4*>(a.grad_slice + (size_t)r * kNA + lane * N) = pk;
    }
}


}  // namespace vaa

extern "C" size_t vaa_loss_rows_ws_bytes(int R);

extern "C" size_t vaa_head_loss_ws_bytes(int R, int V) {
    if (R <= 0 || V <= 0) return 0;
    return vaa::head_ws_slice_offset(R, V) + vaa::head_ws_align((size_t)R * vaa::kNA * sizeof(uint16_t));
}

namespace vaa {

static size_t head_lds_bytes(int nrb) {
    const size_t lds_h = (size_t)kHRing * (nrb * 32 + kHCols) * kHSA * sizeof(uint16_t), lds_t = (size_t)nrb * 32 * kHTileS * sizeof(float);
    return lds_h > lds_t ? lds_h : lds_t;
}

// the LDS a workgroup may ask for: 160 KB on gfx950, the only device this library runs on (vaa_device_check). Round 5 asked the runtime
// (MaxSharedMemoryPerBlock / SharedMemPerBlockOptin) and fell back to 64 KB when the query failed or reported less: a runtime that reports 64 KB
// would then have disabled K3h above 64 rows SILENTLY — and differently on different ranks. With the architectural constant a
// runtime that really refuses the opt-in fails loudly at hipFuncSetAttribute instead.
static size_t device_lds_limit() { return 160 * 1024; }

}  // namespace vaa

extern "C" int vaa_head_loss_rows_applies(int R, int D, int V) {
    if (!(R > 0 && R <= vaa::kHRowsMax && D >= vaa::kHK && (D % vaa::kHK) == 0 && V >= vaa::kA0 + vaa::kNA && (V % 8) == 0 && V <= 131072)) return 0;
    // the weight ring + hidden rows of this row count must fit the device's LDS (128 KB at > 64 rows: a 160 KB-LDS part)
    return vaa::head_lds_bytes(R <= 32 ? 1 : (R <= 64 ? 2 : 4)) <= vaa::device_lds_limit() ? 1 : 0;
}

extern "C" int vaa_head_loss_rows_stats(const uint16_t* hidden, const uint16_t* w_head, int D, const void* rowmap, int R, int B, int L, int V, int mode,
                                        const float* params, void* grad_slice, void* loss_ws, size_t loss_ws_bytes, void* head_ws,
                                        size_t head_ws_bytes, uint16_t* logits_dbg, void* stream) {
    using namespace vaa;
    const char* who = "vaa_head_loss_rows_stats";
    if (!hidden || !w_head || !rowmap || !params || !loss_ws || !head_ws) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (!vaa_head_loss_rows_applies(R, D, V) || B <= 0 || L <= 1 || (long)R > (long)B * (L - 1)) {
        set_error("%s: shape not covered (R=%d <= %d rows, D=%d a multiple of %d, V=%d; B=%d L=%d): use the LM-head GEMM + vaa_loss_rows_stats", who, R,
                  kHRowsMax, D, kHK, V, B, L);
        return VAA_E_UNSUPPORTED;
    }
    if ((((uintptr_t)hidden) | ((uintptr_t)w_head)) & 15u) {  // the rows are fetched in 16-byte pieces (global -> LDS)
        set_error("%s: hidden and w_head must be 16-byte aligned", who);
        return VAA_E_INVALID;
    }
    if (grad_slice && mode != VAA_LOSS_UADA_DDP) {
        set_error("%s: only VAA_LOSS_UADA_DDP has a gradient that does not depend on the folded scalars (mode %d)", who, mode);
        return VAA_E_INVALID;
    }
    if (loss_ws_bytes < vaa_loss_rows_ws_bytes(R) || head_ws_bytes < vaa_head_loss_ws_bytes(R, V)) {
        set_error("%s: workspace too small (loss %zu of %zu B, head %zu of %zu B)", who, loss_ws_bytes, vaa_loss_rows_ws_bytes(R), head_ws_bytes,
                  vaa_head_loss_ws_bytes(R, V));
        return VAA_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    HeadArgs a;
    a.h = hidden; a.w = w_head; a.rowmap = (const int*)rowmap;
    a.nwg = (V + kHCols - 1) / kHCols;
    a.part = (PartStat*)head_ws;
    a.slice_logits = (uint16_t*)((char*)head_ws + head_ws_slice_offset(R, V));
    a.logits_dbg = logits_dbg;
    a.R = R; a.D = D; a.V = V;
    const int nrb = R <= 32 ? 1 : (R <= 64 ? 2 : 4);
    const size_t lds = head_lds_bytes(nrb);
    if (lds > device_lds_limit()) {
        set_error("%s: %zu B of LDS for %d rows exceed this device's %zu B per workgroup (ask vaa_head_loss_rows_applies first)", who, lds, R, device_lds_limit());
        return VAA_E_UNSUPPORTED;
    }
    // the dynamic-LDS opt-in is a property of the kernel function: set once per instantiation and device, not on every call of the step's
    // critical launch path
    static std::atomic<unsigned long long> attr_done[3];  // bit = device ordinal
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int inst = nrb == 1 ? 0 : (nrb == 2 ? 1 : 2);
    const unsigned long long bit = 1ull << (dev & 63);
    if (lds > 64 * 1024 && !(attr_done[inst].load(std::memory_order_acquire) & bit)) {
        const void* fn = nrb == 1 ? (const void*)head_stats_kernel<1> : (nrb == 2 ? (const void*)head_stats_kernel<2> : (const void*)head_stats_kernel<4>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute failed", who);
            return VAA_E_LAUNCH;
        }
        attr_done[inst].fetch_or(bit, std::memory_order_acq_rel);
    }
    const dim3 grid((unsigned)a.nwg), blk(kHT);
    if (nrb == 1) VAA_LAUNCH((head_stats_kernel<1>), grid, blk, lds, st, a);
    else if (nrb == 2) VAA_LAUNCH((head_stats_kernel<2>), grid, blk, lds, st, a);
    else VAA_LAUNCH((head_stats_kernel<4>), grid, blk, lds, st, a);
    int rc = check_launch(who);
    if (rc != VAA_OK) return rc;
    HeadFinishArgs f;
    f.part_in = a.part; f.slice_logits = a.slice_logits; f.rowmap = a.rowmap;
    f.part_out = (PartStat*)loss_ws;
    f.slice_out = (SliceStat*)((char*)loss_ws + (size_t)R * 4 * sizeof(PartStat));
    f.grad_slice = (uint16_t*)grad_slice;
    f.R = R; f.nwg = a.nwg; f.split = rows_split(R, V); f.mode = mode; f.w = params[0];
    VAA_LAUNCH(head_finish_kernel, dim3((unsigned)R), dim3(256), 0, st, f);
    return check_launch(who);
}
