// head_fused_probe.hip — PROBE (round 4; measured, NOT adopted: profiles/r04_head_fused.txt): SURVEY.md section 8f-2 as the
// survey wrote it, the LM head on the labelled rows FUSED with K3's statistics. Not part of libvaa_hip.so; tools/head_bench.py builds it into a
// variant library (tools/scratch/build_variant.sh HEAD tools/probe/head_fused_probe.hip) and binds its three entry points itself.
//
// Replaces `logits = lm_head(hidden_rows)` (modeling_prismatic.py:404-415 -> HF Llama's lm_head, bf16) + the statistics pass of K3
// (vaa_loss_rows_stats: HF's CE terms and weighted_loss's soft-argmax, UADA_ddp.py:99-124) for the data-parallel UADA step, whose gradient
// lives in the 256 action columns: the [R',V] logits are never written to memory.
//
//   head_stats_kernel<NQ>   grid = ceil(V / 128) workgroups of 8 waves; workgroup w owns the 128 vocabulary columns [128 w, 128 w + 128):
//       * the 128 weight rows W[n, :] (8 KB each, contiguous in K) are streamed from HBM exactly once, straight into MFMA B fragments
//         (lane = (column n, k-group): 16 bytes per lane, a row's four lanes cover 64 contiguous bytes), three 128-wide k-chunks ahead
//         of their use — 263 MB at V = 32,064, D = 4,096: the kernel is a weight stream, its floor is 263 MB / HBM bandwidth;
//       * the hidden rows H [R' <= 128, D] (1 MB, L2-resident, read by every workgroup) go through a double-buffered LDS image in full
//         256-byte row pieces, one barrier per 128-wide k-chunk; every wave multiplies all R' rows against its own 16 columns
//         (mfma_f32_16x16x32_bf16; 32 % of the matrix pipe suffices to keep up with the stream);
//       * epilogue: the accumulators are rounded to bf16 (what the reference's bf16 head hands to `.float()`), laid out as a [R', 128] tile
//         in LDS and reduced per row to {max, sum exp, argmax, label logit} = one PartStat per (row, workgroup); the two workgroups that
//         own the action columns 31744..31999 also leave those logits in a [R', 256] fp32 buffer.
//   head_finish_kernel      grid = R' workgroups: folds a row's ceil(V / 128) PartStats into ONE (stored in K3's workspace layout, the other
//       parts neutral), computes the action-slice statistics with the arithmetic of rows_stats_kernel (same bits for the same logits) and
//       — UADA_DDP — writes the gradient slice. vaa_step_epilogue then folds the rows exactly as it does behind vaa_loss_rows_stats.
#include "vaa_common.h"
#include "vaa_rows.h"

namespace vaa {

typedef short v8s_h __attribute__((ext_vector_type(8)));
typedef float v4f_h __attribute__((ext_vector_type(4)));

constexpr int kHT = 256;             // threads per workgroup: 4 waves, ONE per SIMD (up to 512 VGPRs each: the weight stream lives in registers)
constexpr int kHCols = 128;          // vocabulary columns per workgroup: 32 per wave
constexpr int kHK = 128;             // k-chunk
constexpr int kHSA = kHK + 8;        // padded LDS row of the H image (bf16 elements): 272 B, conflict-free 16-byte fragment reads
constexpr int kHTileS = kHCols + 4;  // padded row of the fp32 logits tile
constexpr int kHRowsMax = 128;
constexpr int kHRing = 4;            // LDS ring of H chunks
#ifdef VAA_HEAD_PLAIN_LOADS
#define VAA_HEAD_LOAD(p) (*(p))
#else
#define VAA_HEAD_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifndef VAA_HEAD_SETS1
#define VAA_HEAD_SETS1 8
#endif
#ifndef VAA_HEAD_SETS4
#define VAA_HEAD_SETS4 4
#endif
// weight register sets (32 VGPRs each; SETS - 1 k-chunks are in flight ahead of the one being multiplied), by the hidden rows' register needs
template <int NRB> struct HeadSets { static constexpr int value = NRB >= 4 ? VAA_HEAD_SETS4 : VAA_HEAD_SETS1; };

struct HeadArgs {
    const uint16_t* h;      // [R, D] bf16 hidden rows (final norm applied)
    const uint16_t* w;      // [V, D] bf16 LM-head weight
    const int* rowmap;      // K3's row map {R, #action rows, 0, 0} + RowMap[R]
    PartStat* part;         // [R][nwg]
    float* slice_logits;    // [R][256] action-column logits (bf16-rounded values)
    uint16_t* logits_dbg;   // [R][V] bf16 or nullptr (tests)
    int R, D, V, nwg;
};

// NRB = 32-row blocks of hidden rows (R' <= 32 NRB). A wave owns TWO 16-column blocks and all rows: mfma_f32_16x16x32_bf16 with the weight
// rows as B fragments straight from global memory (lane = (column, k-group): 16 bytes per lane, the four lanes of a column cover 64 contiguous
// bytes — the widest piece of one row a single load instruction can hand to MFMA operands; a per-lane-row layout, 64 lines per instruction,
// measured 2.4x slower) and the hidden rows as A fragments from LDS, each read feeding two MFMAs.
template <int NRB>
__global__ __launch_bounds__(kHT) void head_stats_kernel(HeadArgs a) {
    extern __shared__ __align__(16) unsigned char head_smem[];
    uint16_t* hb = reinterpret_cast<uint16_t*>(head_smem);  // [kHRing][ROWS][kHSA]
    float* tile = reinterpret_cast<float*>(head_smem);      // epilogue: [ROWS][kHTileS]
    constexpr int ROWS = NRB * 32, NQ = NRB * 2;            // NQ: 16-row blocks
    constexpr int HL = ROWS * (kHK / 8) / kHT;  // 16-byte loads per thread and H chunk: 2 NRB
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * kHCols;
    const int nchunks = a.D / kHK;
#ifndef VAA_HEAD_STAGGER
#define VAA_HEAD_STAGGER 5
#endif
    // k-chunk order: workgroup w starts at chunk (5 w) mod nchunks and wraps around. Without the stagger every workgroup reads the SAME k offset of
    // its 128 weight rows (8 KB apart) at the same time: the requests of the whole chip fall on a few memory channels.
    const int kstart = (int)((blockIdx.x * (unsigned)VAA_HEAD_STAGGER) % (unsigned)nchunks);
    auto kchunk = [&](int ch) { const int cc = min(ch, nchunks - 1) + kstart; return cc >= nchunks ? cc - nchunks : cc; };
    // this lane's two weight rows: columns n0 + 32 wv + 16 cb + c (columns beyond V re-read the last row; their results are never used)
    const uint16_t* wrow[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) wrow[cb] = a.w + (size_t)min(n0 + wv * 32 + cb * 16 + c, a.V - 1) * a.D + g * 16;
    constexpr int SETS = HeadSets<NRB>::value, DEPTH = SETS - 1;
    static_assert(SETS >= 4 && SETS % 2 == 0, "the rotation pairs the weight sets with the two H register sets");
    // k-slices: within a 128-wide chunk (two 128-byte lines of a weight row) k-group g owns the 32-byte slot g of each line; MFMA 2 L + p takes
    // the p-th 16 bytes of its slot in line L. The two load instructions of a line touch the SAME 16 lines (one per column) instead of two
    // different half-lines each (any assignment of k to lanes is valid as long as the A and the B fragments agree)
    auto kofs = [](int jj) { return (jj >> 1) * 64 + (jj & 1) * 8; };
    v8s_h wreg[SETS][8];  // [set][cb * 4 + j]
    auto load_w = [&](v8s_h (&dst)[8], int ch) {  // unconditional (the tail re-requests the last chunk): the compiler counts the requests
        const size_t off = (size_t)kchunk(ch) * kHK;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)  // streamed once: nontemporal, kept out of the L2 the H rows live in
                dst[cb * 4 + jj] = VAA_HEAD_LOAD(reinterpret_cast<const v8s_h*>(wrow[cb] + off + kofs(jj)));
    };
    uint4 hreg[2][HL];
    auto load_h = [&](uint4 (&dst)[HL], int ch) {
        const int chc = kchunk(ch);
#pragma unroll
        for (int it = 0; it < HL; ++it) {
            const int idx = tid + it * kHT, row = idx >> 4, piece = idx & 15;
            dst[it] = make_uint4(0, 0, 0, 0);
            if (row < a.R) dst[it] = *reinterpret_cast<const uint4*>(a.h + (size_t)row * a.D + chc * kHK + piece * 8);
        }
    };
    auto store_h = [&](const uint4 (&src)[HL], int buf) {
#pragma unroll
        for (int it = 0; it < HL; ++it) {
            const int idx = tid + it * kHT, row = idx >> 4, piece = idx & 15;
            *reinterpret_cast<uint4*>(&hb[(buf * ROWS + row) * kHSA + piece * 8]) = src[it];
        }
    };
    v4f_h acc[2][NQ];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[cb][q] = (v4f_h){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: H chunks 0..2 into the ring, weight chunks 0..2 in flight, H chunk 3 requested ----
    // Request ORDER is the design: loads complete in order, so waiting for an H chunk drains every weight request issued before it. H chunk
    // x is therefore requested a full step before weight chunk x - 1: the wait for it (end of step x - 3) leaves two weight chunks in flight.
    load_h(hreg[0], 0);
    store_h(hreg[0], 0);
    load_h(hreg[0], 1);
    load_h(hreg[1], 2);
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_w(wreg[s], s);
    store_h(hreg[0], 1);
    store_h(hreg[1], 2);
    load_h(hreg[1], 3);
    __syncthreads();

    auto step = [&](int ch, const v8s_h (&wcur)[8], v8s_h (&wnext)[8], uint4 (&hnew)[HL], const uint4 (&hold)[HL]) {
        load_h(hnew, ch + 4);
        load_w(wnext, ch + DEPTH);
        const uint16_t* ap = &hb[((ch & (kHRing - 1)) * ROWS + c) * kHSA + g * 16];
        v8s_h af[2][NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) af[0][q] = *reinterpret_cast<const v8s_h*>(ap + q * 16 * kHSA + kofs(0));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj + 1 < 4) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) af[(jj + 1) & 1][q] = *reinterpret_cast<const v8s_h*>(ap + q * 16 * kHSA + kofs(jj + 1));
            }
            __builtin_amdgcn_sched_barrier(0);  // the next k-step's LDS reads are issued before this step's MFMAs
            if (ch < nchunks) {  // workgroup-uniform (false only in the padded steps behind the last chunk)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc[cb][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[jj & 1][q], wcur[cb * 4 + jj], acc[cb][q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        store_h(hold, (ch + 3) & (kHRing - 1));  // H chunk ch + 3 (requested a step ago): its slot was last read in step ch - 1
        __syncthreads();
    };
    static_assert(kHRing == 4, "H chunk x is requested in step x - 4 and stored in step x - 3");
    // SETS steps per iteration, fully unrolled: register sets are indexed statically and the request counts stay static. Steps beyond the last
    // chunk (D / 128 not a multiple of SETS) only re-request the last chunk (unconditional loads) and skip the MFMAs.
    for (int ch0 = 0; ch0 < nchunks; ch0 += SETS) {
#pragma unroll
        for (int u = 0; u < SETS; ++u) step(ch0 + u, wreg[u], wreg[(u + DEPTH) % SETS], hreg[u & 1], hreg[(u + 1) & 1]);
    }
    // (the loop's last barrier: every wave is done with the H ring — the LDS becomes the logits tile)

    // ---- epilogue: bf16-rounded logits -> LDS tile [ROWS][128] ----
    // C/D layout: column = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(q * 16 + g * 4 + r) * kHTileS + wv * 32 + cb * 16 + c] = bf16_bits_to_f32(f32_to_bf16_bits(acc[cb][q][r]));
    __syncthreads();
    const int ncols = min(kHCols, a.V - n0);
    // per-row statistics: 2 threads per row, thread `part` takes the columns part, part + 2, ...
    {
        const int row = tid >> 1, part = tid & 1;
        const bool live = row < ROWS && row < a.R;
        float m = -INFINITY;
        int mi = 0x7fffffff;
        if (live) {
            const float* tr = tile + row * kHTileS;
            for (int cl = part; cl < ncols; cl += 2) {  // increasing columns: the first maximum wins
                const float v = tr[cl];
                if (v > m) { m = v; mi = n0 + cl; }
            }
        }
        {
            const float om = __shfl_xor(m, 1, 64);
            const int oi = __shfl_xor(mi, 1, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float s = 0.0f;
        if (live) {
            const float* tr = tile + row * kHTileS;
            for (int cl = part; cl < ncols; cl += 2) s += expf(tr[cl] - m);
        }
        s += __shfl_xor(s, 1, 64);
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
            a.slice_logits[(size_t)r * kNA + (n0 - kA0) + cl] = tile[r * kHTileS + cl];
        }
    }
    if (a.logits_dbg) {
        for (int idx = tid; idx < rows_live * kHCols; idx += kHT) {
            const int r = idx >> 7, cl = idx & 127;
            if (cl < ncols) a.logits_dbg[(size_t)r * a.V + n0 + cl] = (uint16_t)f32_to_bf16_bits(tile[r * kHTileS + cl]);
        }
    }
}

struct HeadFinishArgs {
    const PartStat* part_in;    // [R][nwg]
    const float* slice_logits;  // [R][256]
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
        const float4 v0 = *reinterpret_cast<const float4*>(a.slice_logits + (size_t)r * kNA + lane * N);
        const float4 v1 = *reinterpret_cast<const float4*>(a.slice_logits + (size_t)r * kNA + lane * N + 4);
        x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
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

static size_t head_align(size_t n) { return (n + 255) / 256 * 256; }

}  // namespace vaa

extern "C" size_t vaa_loss_rows_ws_bytes(int R);

extern "C" size_t vaa_head_loss_ws_bytes(int R, int V) {
    if (R <= 0 || V <= 0) return 0;
    const size_t nwg = (size_t)(V + vaa::kHCols - 1) / vaa::kHCols;
    return vaa::head_align((size_t)R * nwg * sizeof(vaa::PartStat)) + vaa::head_align((size_t)R * vaa::kNA * sizeof(float));
}

extern "C" int vaa_head_loss_rows_applies(int R, int D, int V) {
    return (R > 0 && R <= vaa::kHRowsMax && D >= vaa::kHK && (D % vaa::kHK) == 0 && V >= vaa::kA0 + vaa::kNA && (V % 8) == 0 && V <= 131072) ? 1 : 0;
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
    a.slice_logits = (float*)((char*)head_ws + head_align((size_t)R * a.nwg * sizeof(PartStat)));
    a.logits_dbg = logits_dbg;
    a.R = R; a.D = D; a.V = V;
    const int nrb = R <= 32 ? 1 : (R <= 64 ? 2 : 4);
    const size_t lds_h = (size_t)kHRing * nrb * 32 * kHSA * sizeof(uint16_t), lds_t = (size_t)nrb * 32 * kHTileS * sizeof(float);
    const size_t lds = lds_h > lds_t ? lds_h : lds_t;
    const void* fn = nrb == 1 ? (const void*)head_stats_kernel<1> : (nrb == 2 ? (const void*)head_stats_kernel<2> : (const void*)head_stats_kernel<4>);
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("%s: hipFuncSetAttribute failed", who);
        return VAA_E_LAUNCH;
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
