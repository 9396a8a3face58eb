// vaa_attention.hip — OPTIONAL model-side operator (include/vaa_model_ops.h): softmax attention forward + backward on the
// gfx950 matrix cores for the short sequences of the OpenVLA step (T = 256..~330 tokens, head dim 64 / 72 / 128).
//
// Everything is formulated TRANSPOSED so that no operand ever has to be re-laid-out in registers:
//   forward   S^T = K Q^T            A = K rows from LDS (16 B reads), B = Q^T held in registers (16 B global reads of Q rows)
//             O^T += V^T P^T         A = V^T through ds_read_b64_tr_b16 (hardware transpose read of the row-major V tile),
//                                    B = P^T = exp2(S^T - m) straight from the S^T accumulators
// In the C/D layout of mfma_f32_16x16x32_bf16 (col = lane&15, row = 4*(lane>>4)+reg) a lane's column is its QUERY, so the
// online-softmax state (running max, running sum, rescale factor) is one scalar per lane, and the 8 accumulator values of two
// 16-key tiles are exactly the 8 k-slots of that lane's B operand for the next MFMA (k-slot j<4 <-> key 4g+j, j>=4 <-> key
// 16+4g+(j-4); the A operand is fetched with the same permutation, which the transpose read delivers for free).
// The backward (dq kernel, dk/dv kernel) uses the same two access patterns; see the kernels.
//
// One workgroup = 4 waves = 64 queries (forward, dq) or 64 keys (dk/dv) of one (batch, head); K/V (or Q/dO) stream through
// LDS in 64-row tiles, register-staged so the next tile's global loads are in flight during the MFMAs. Workgroups of the same
// (batch, head) are placed on the same XCD (blockIdx % 8) so that its K/V stay in that XCD's L2.
#include "vaa_common.h"

#include <cstdlib>
#include <mutex>

#include "../../include/vaa_model_ops.h"

namespace vaa {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;

struct AttnStr {
    long b, t, h;  // element strides of a [B,T,H,hd] view (last dim contiguous)
};

struct AttnFwdArgs {
    const uint16_t *q, *k, *v;
    uint16_t* o;
    float* lse;  // [B,H,T] natural-log logsumexp of scale*q.k over the visible keys
    AttnStr sq, sk, sv, so;
    const int32_t* cu;  // optional [B+1] cumulative token counts: sequences are packed back to back along the token axis (batch stride
                        // unused), sample b has cu[b+1]-cu[b] tokens; T is then the MAXIMUM length (grid size, lse row stride)
    int B, H, T, hd;
    float scale_log2;  // softmax scale * log2(e)
};

struct SeqInfo {
    int T;      // tokens of this sample
    long tok0;  // first token of this sample on the packed token axis (0 when not packed)
    bool packed;
};
__device__ __forceinline__ SeqInfo seq_info(const int32_t* cu, int b, int Tmax) {
    SeqInfo si;
    si.packed = cu != nullptr;
    si.tok0 = si.packed ? cu[b] : 0;
    si.T = si.packed ? cu[b + 1] - cu[b] : Tmax;
    return si;
}
__device__ __forceinline__ long row_base(const AttnStr& st, const SeqInfo& si, int b, int h) {
    return (si.packed ? si.tok0 * st.t : (long)b * st.b) + (long)h * st.h;
}

constexpr int kTile = 64;  // rows (keys or queries) per LDS tile = queries per workgroup

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ v4s lds_tr16(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(p));
}

__device__ __forceinline__ v8s cat8(v4s lo, v4s hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair (v_cvt_pk_bf16_f32, round-to-nearest-even)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const bf2 r = __builtin_convertvector((f2){lo, hi}, bf2);
    return *reinterpret_cast<const uint32_t*>(&r);
}

__device__ __forceinline__ v8s pack8(v4f a, v4f b) {
    uint4 r;
    r.x = cvt_pk_bf16(a[0], a[1]);
    r.y = cvt_pk_bf16(a[2], a[3]);
    r.z = cvt_pk_bf16(b[0], b[1]);
    r.w = cvt_pk_bf16(b[2], b[3]);
    return *reinterpret_cast<v8s*>(&r);
}

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// Raw buffer descriptor over the rows [0,T) of one (batch, head) slice: loads past the last valid element return 0, so tile
// rows >= T need no branch. (Columns hd..HDP of a padded head dim read the neighbouring head's finite values; every product
// they enter has a zero-padded register operand on the other side, and output rows >= hd are never stored.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const uint16_t* base, long st, int T, int hd) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((((long)T - 1) * st + hd) * 2), 0x00020000);
}

// Register-staged copy of a [kTile x HDP] tile of a [*,T,*,hd] tensor.
template <int HDP>
struct TileRegs {
    static constexpr int kChunks = kTile * (HDP / 8) / 256;  // 16-byte chunks per thread
    v4u r[kChunks];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int st_bytes, int row0) {
#pragma unroll
        for (int it = 0; it < kChunks; ++it) {
            const int ch = threadIdx.x + it * 256;
            const int row = ch / (HDP / 8), cc = ch % (HDP / 8);
            r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (row0 + row) * st_bytes + cc * 16, 0, 0);
        }
    }
    __device__ __forceinline__ void store(uint16_t* lds, int stride) const {
#pragma unroll
        for (int it = 0; it < kChunks; ++it) {
            const int ch = threadIdx.x + it * 256;
            const int row = ch / (HDP / 8), cc = ch % (HDP / 8);
            *reinterpret_cast<v4u*>(lds + row * stride + cc * 8) = r[it];
        }
    }
};

// blockIdx -> (batch*head pair, 64-row block); all blocks of a pair share blockIdx % 8 (one XCD, one L2); heavy (late) causal
// blocks first.
__device__ __forceinline__ bool block_to_pair(int nblk, int npairs, int& pair, int& blk, bool late_first = true) {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    pair = (slot / nblk) * 8 + xcd;
    blk = late_first ? nblk - 1 - (slot % nblk) : slot % nblk;
    return pair < npairs;
}

// ---------------------------------------------------------------- forward ----------------------------------------------------------------
// G = 16-query column groups per wave (1 or 2): with G = 2 a workgroup covers 128 queries and every K / V fragment read from LDS feeds two
// MFMAs (one per group), i.e. half the LDS traffic and half the tile loads per unit of work; per query the arithmetic and its order are
// those of G = 1 (same bits).
template <int KS, int NT, bool CAUSAL, int G>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnFwdArgs a) {
    constexpr int HDP = KS * 32;
    constexpr int SK = HDP + 8;   // K tile row stride (bf16): dword stride = 4 mod 8 -> conflict-free 16-byte row reads
    constexpr int SV = HDP + 16;  // V tile row stride: dword stride = 8 mod 16 -> conflict-free transpose reads
    constexpr int QB = kTile * G, QW = 16 * G;  // queries per workgroup / per wave
    __shared__ __attribute__((aligned(16))) uint16_t sK[kTile * SK];
    __shared__ __attribute__((aligned(16))) uint16_t sV[kTile * SV];

    const int nqb = (a.T + QB - 1) / QB;
    int pair, qb;
    if (!block_to_pair(nqb, a.B * a.H, pair, qb)) return;
    const int b = pair / a.H, h = pair - b * a.H;
    const SeqInfo si = seq_info(a.cu, b, a.T);
    const int T = si.T;
    if (qb * QB >= T) return;  // block beyond this sample's length (packed batches)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = qb * QB + wv * QW;  // this wave's first query; group j holds queries q0 + 16 j + [0, 16)

    const uint16_t* qp = a.q + row_base(a.sq, si, b, h);
    const uint16_t* kp = a.k + row_base(a.sk, si, b, h);
    const uint16_t* vp = a.v + row_base(a.sv, si, b, h);

    // Q^T fragments: lane (c,g), group j, k-step ks <-> Q[q0 + 16 j + c][32 ks + 8 g .. +8]
    v8s qf[G][KS];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = q0 + 16 * j + c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            uint4 r = make_uint4(0, 0, 0, 0);
            if (q < T && d0 < a.hd) r = *reinterpret_cast<const uint4*>(qp + (long)q * a.sq.t + d0);
            qf[j][ks] = *reinterpret_cast<v8s*>(&r);
        }
    }

    const int kend = CAUSAL ? min(T, (qb + 1) * QB) : T;  // keys [0, kend) are visible to this workgroup
    const int ntile = (kend + kTile - 1) / kTile;

    v4f acc[G][NT];
    float m[G], lsum[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[j][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        m[j] = -INFINITY;
        lsum[j] = 0.0f;
    }

    const __amdgpu_buffer_rsrc_t krs = slice_rsrc(kp, a.sk.t, T, a.hd), vrs = slice_rsrc(vp, a.sv.t, T, a.hd);
    const int kst = (int)a.sk.t * 2, vst = (int)a.sv.t * 2;
    TileRegs<HDP> rk, rv;
    rk.load(krs, kst, 0);
    rv.load(vrs, vst, 0);
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();  // every wave is done reading the previous tile
        rk.store(sK, SK);
        rv.store(sV, SV);
        __syncthreads();
        if (kt + 1 < ntile) {  // next tile's loads fly during the MFMAs
            rk.load(krs, kst, (kt + 1) * kTile);
            rv.load(vrs, vst, (kt + 1) * kTile);
        }
        const int key0 = kt * kTile;
        if (CAUSAL && key0 > q0 + QW - 1) continue;  // nothing visible to this wave in this tile (wave-uniform)

        v4f st[G][4];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {  // all fragment reads of two 16-key tiles first, then their MFMAs
            v8s kfr[2][KS];
#pragma unroll
            for (int rl = 0; rl < 2; ++rl)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kfr[rl][ks] = *reinterpret_cast<const v8s*>(&sK[((2 * rp + rl) * 16 + c) * SK + ks * 32 + g * 8]);
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int rl = 0; rl < 2; ++rl) {
                    v4f t = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[rl][ks], qf[j][ks], t, 0, 0, 0);
                    st[j][2 * rp + rl] = t;
                }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int qj0 = q0 + 16 * j, q = qj0 + c;
            const bool need_mask = (key0 + kTile > T) || (CAUSAL && key0 + kTile - 1 > qj0);  // wave-uniform
            if (need_mask) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + rt * 16 + g * 4 + r;
                        if (key >= T || (CAUSAL && key > q)) st[j][rt][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[j][rt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[j], mx * a.scale_log2);  // scale > 0: max commutes with the scaling
            const float mu = (mn == -INFINITY) ? 0.0f : mn;
            const float alpha = fast_exp2(m[j] - mu);  // m = -inf -> 0
            m[j] = mn;
            float ps = 0.0f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(__builtin_fmaf(st[j][rt][r], a.scale_log2, -mu));
                    st[j][rt][r] = p;
                    ps += p;
                }
            lsum[j] = lsum[j] * alpha + ps;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[j][nt] *= alpha;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint16_t* vrow = &sV[(half * 32 + 4 * g + (c >> 2)) * SV + (c & 3) * 4];
            v8s vfr[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) vfr[nt] = cat8(lds_tr16(vrow + nt * 16), lds_tr16(vrow + 16 * SV + nt * 16));
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const v8s pf = pack8(st[j][2 * half], st[j][2 * half + 1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[nt], pf, acc[j][nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = q0 + 16 * j + c;
        float ls = lsum[j];
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        if (q < T) {
            const float inv = ls > 0.0f ? 1.0f / ls : 0.0f;
            uint16_t* op = a.o + row_base(a.so, si, b, h) + (long)q * a.so.t;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int d0 = nt * 16 + g * 4;
                if (d0 < a.hd) {
                    uint2 w;
                    w.x = cvt_pk_bf16(acc[j][nt][0] * inv, acc[j][nt][1] * inv);
                    w.y = cvt_pk_bf16(acc[j][nt][2] * inv, acc[j][nt][3] * inv);
                    *reinterpret_cast<uint2*>(op + d0) = w;
                }
            }
            if (g == 0) a.lse[((long)b * a.H + h) * a.T + q] = (m[j] + __builtin_log2f(ls)) * 0.69314718055994531f;
        }
    }
}

// ---------------------------------------------------------------- backward ---------------------------------------------------------------
struct AttnBwdArgs {
    const uint16_t *q, *k, *v, *o, *dout;
    const float* lse;  // [B,H,T] from the forward
    float* dsum;       // [B,H,T] workspace: D = rowsum(dO * O), written by the dq kernel, read by the dk/dv kernel
    uint16_t *dq, *dk, *dv;
    AttnStr sq, sk, sv, so, sdo, sdq, sdk, sdv;
    const int32_t* cu;  // as in AttnFwdArgs
    const float *rope_cos, *rope_sin;  // optional float32 [T, hd/2]: q and k were rotated before the forward; dq/dk are returned
                                       // w.r.t. the UN-rotated projections (adjoint rotation fused into the epilogues)
    int B, H, T, hd;
    float scale, scale_log2;
};

// Writes one key's / query's gradient row from transposed accumulators (lane: column = that row, rows = head-dim 16 nt + 4 g + r),
// scaled, optionally through the adjoint of HF's rotate_half rotary embedding: g1' = g1 c + g2 s, g2' = g2 c - g1 s.
template <int NT>
__device__ __forceinline__ void store_grad_row(uint16_t* op, const v4f (&acc)[NT], float scale, int g, int hd, const float* cosr, const float* sinr) {
    if (cosr) {  // host guarantees hd == 16 NT here
#pragma unroll
        for (int nt = 0; nt < NT / 2; ++nt) {
            const int d0 = nt * 16 + g * 4;
            const float4 c4 = *reinterpret_cast<const float4*>(cosr + d0), s4 = *reinterpret_cast<const float4*>(sinr + d0);
            const float cs[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
            float o1[4], o2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g1 = acc[nt][r] * scale, g2 = acc[nt + NT / 2][r] * scale;
                o1[r] = g1 * cs[r] + g2 * sn[r];
                o2[r] = g2 * cs[r] - g1 * sn[r];
            }
            uint2 w;
            w.x = cvt_pk_bf16(o1[0], o1[1]); w.y = cvt_pk_bf16(o1[2], o1[3]);
            *reinterpret_cast<uint2*>(op + d0) = w;
            w.x = cvt_pk_bf16(o2[0], o2[1]); w.y = cvt_pk_bf16(o2[2], o2[3]);
            *reinterpret_cast<uint2*>(op + d0 + NT * 8) = w;
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int d0 = nt * 16 + g * 4;
        if (d0 < hd) {
            uint2 w;
            w.x = cvt_pk_bf16(acc[nt][0] * scale, acc[nt][1] * scale);
            w.y = cvt_pk_bf16(acc[nt][2] * scale, acc[nt][3] * scale);
            *reinterpret_cast<uint2*>(op + d0) = w;
        }
    }
}

// B-operand fragments of a row-major [*, hd] matrix row: lane (c,g), k-step ks <-> row[32 ks + 8 g .. +8] (zero outside).
template <int KS>
__device__ __forceinline__ void load_row_frags(v8s (&f)[KS], const uint16_t* row, bool valid, int g, int hd) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 32 + g * 8;
        uint4 r = make_uint4(0, 0, 0, 0);
        if (valid && d0 < hd) r = *reinterpret_cast<const uint4*>(row + d0);
        f[ks] = *reinterpret_cast<v8s*>(&r);
    }
}

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// A-operand fragments of rows [rt*16, rt*16+16) of an LDS tile (lane (c,g), k-step ks <-> row rt*16+c, columns 32 ks + 8 g .. +8)
template <int KS>
__device__ __forceinline__ void load_tile_frags(v8s (&af)[KS], const uint16_t* tile, int stride, int rt, int c, int g) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const v8s*>(&tile[(rt * 16 + c) * stride + ks * 32 + g * 8]);
}
template <int KS>
__device__ __forceinline__ v4f frag_dot(const v8s (&af)[KS], const v8s (&bf)[KS]) {
    v4f acc = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bf[ks], acc, 0, 0, 0);
    return acc;
}

// dQ^T = sum over key tiles of K^T dS^T, with S^T = K Q^T and dP^T = V dO^T recomputed; dS^T = P^T o (dP^T - D). One workgroup =
// 64 G queries (G column groups of 16 per wave: every K / V fragment read feeds G MFMAs); a lane's column is its query, so lse and D are
// per-lane scalars. Also produces D for the dk/dv kernel.
template <int KS, int NT, bool CAUSAL, int G>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs a) {
    constexpr int HDP = KS * 32;
    constexpr int SK = HDP + 16, SV = HDP + 8;  // K is read by rows (16) AND transposed (32 reads): its stride favours the transpose reads
    constexpr int QB = kTile * G, QW = 16 * G;
    __shared__ __attribute__((aligned(16))) uint16_t sK[kTile * SK];
    __shared__ __attribute__((aligned(16))) uint16_t sV[kTile * SV];

    const int nqb = (a.T + QB - 1) / QB;
    int pair, qb;
    if (!block_to_pair(nqb, a.B * a.H, pair, qb)) return;
    const int b = pair / a.H, h = pair - b * a.H;
    const SeqInfo si = seq_info(a.cu, b, a.T);
    const int T = si.T;
    if (qb * QB >= T) return;  // block beyond this sample's length (packed batches)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = qb * QB + wv * QW;

    const uint16_t* kp = a.k + row_base(a.sk, si, b, h);
    const uint16_t* vp = a.v + row_base(a.sv, si, b, h);
    v8s qf[G][KS], dof[G][KS];
    float Dq[G], lse2[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = q0 + 16 * j + c;
        const bool qv = q < T;
        load_row_frags<KS>(qf[j], a.q + row_base(a.sq, si, b, h) + (long)q * a.sq.t, qv, g, a.hd);
        load_row_frags<KS>(dof[j], a.dout + row_base(a.sdo, si, b, h) + (long)q * a.sdo.t, qv, g, a.hd);
        float dq_ = 0.0f;
        v8s of[KS];
        load_row_frags<KS>(of, a.o + row_base(a.so, si, b, h) + (long)q * a.so.t, qv, g, a.hd);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dq_ += bf16_bits_to_f32((uint16_t)of[ks][e]) * bf16_bits_to_f32((uint16_t)dof[j][ks][e]);
        dq_ += __shfl_xor(dq_, 16, 64);
        dq_ += __shfl_xor(dq_, 32, 64);
        Dq[j] = dq_;
        const long rowid = ((long)b * a.H + h) * a.T + q;
        if (qv && g == 0) a.dsum[rowid] = dq_;
        lse2[j] = qv ? a.lse[rowid] * 1.4426950408889634f : INFINITY;  // invalid query -> P = 0
    }

    const int kend = CAUSAL ? min(T, (qb + 1) * QB) : T;
    const int ntile = (kend + kTile - 1) / kTile;
    v4f acc[G][NT];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[j][nt] = (v4f){0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t krs = slice_rsrc(kp, a.sk.t, T, a.hd), vrs = slice_rsrc(vp, a.sv.t, T, a.hd);
    const int kst = (int)a.sk.t * 2, vst = (int)a.sv.t * 2;
    TileRegs<HDP> rk, rv;
    rk.load(krs, kst, 0);
    rv.load(vrs, vst, 0);
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        rk.store(sK, SK);
        rv.store(sV, SV);
        __syncthreads();
        if (kt + 1 < ntile) {
            rk.load(krs, kst, (kt + 1) * kTile);
            rv.load(vrs, vst, (kt + 1) * kTile);
        }
        const int key0 = kt * kTile;
        if (CAUSAL && key0 > q0 + QW - 1) continue;
        const bool need_mask = (key0 + kTile > T) || (CAUSAL && key0 + kTile - 1 > q0);  // wave-uniform (q0 = the wave's smallest query)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (CAUSAL && key0 + half * 32 > q0 + QW - 1) continue;
            v4f ds[G][2];
#pragma unroll
            for (int rl = 0; rl < 2; ++rl) {
                const int rt = 2 * half + rl;
                v8s ak[KS], av[KS];
                load_tile_frags<KS>(ak, sK, SK, rt, c, g);
                load_tile_frags<KS>(av, sV, SV, rt, c, g);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const v4f st = frag_dot<KS>(ak, qf[j]);
                    const v4f dp = frag_dot<KS>(av, dof[j]);
                    const int q = q0 + 16 * j + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = fast_exp2(__builtin_fmaf(st[r], a.scale_log2, -lse2[j]));
                        if (need_mask) {
                            const int key = key0 + rt * 16 + g * 4 + r;
                            if (key >= T || (CAUSAL && key > q)) p = 0.0f;
                        }
                        ds[j][rl][r] = p * (dp[r] - Dq[j]);
                    }
                }
            }
            const uint16_t* krow = &sK[(half * 32 + 4 * g + (c >> 2)) * SK + (c & 3) * 4];
            v8s kfr[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) kfr[nt] = cat8(lds_tr16(krow + nt * 16), lds_tr16(krow + 16 * SK + nt * 16));
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const v8s dsf = pack8(ds[j][0], ds[j][1]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[nt], dsf, acc[j][nt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = q0 + 16 * j + c;
        if (q < T) {
            const long ro = (si.tok0 + q) * (a.hd >> 1);
            store_grad_row<NT>(a.dq + row_base(a.sdq, si, b, h) + (long)q * a.sdq.t, acc[j], a.scale, g, a.hd,
                               a.rope_cos ? a.rope_cos + ro : nullptr, a.rope_cos ? a.rope_sin + ro : nullptr);
        }
    }
}

// dK^T += Q^T dS, dV^T += dO^T P over query tiles, with S = Q K^T and dP = dO V^T recomputed (A = Q / dO rows from LDS, B = K^T / V^T
// fragments of this wave's 16 G keys held in registers: every Q / dO fragment read feeds G MFMAs): a lane's column is its KEY, rows are
// queries, so lse and D come from LDS.
template <int KS, int NT, bool CAUSAL, int G>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs a) {
    constexpr int HDP = KS * 32;
    constexpr int SQ = HDP + 16;  // both tiles are read by rows AND transposed; the transpose reads are twice as many -> their stride
    constexpr int KB = kTile * G, KW = 16 * G;  // keys per workgroup / per wave
    __shared__ __attribute__((aligned(16))) uint16_t sQ[kTile * SQ];
    __shared__ __attribute__((aligned(16))) uint16_t sDO[kTile * SQ];
    __shared__ float sL[kTile], sD[kTile];

    const int nkb = (a.T + KB - 1) / KB;
    int pair, kb;
    if (!block_to_pair(nkb, a.B * a.H, pair, kb, !CAUSAL)) return;  // causal: early key blocks see the most queries -> first
    const int b = pair / a.H, h = pair - b * a.H;
    const SeqInfo si = seq_info(a.cu, b, a.T);
    const int T = si.T;
    if (kb * KB >= T) return;  // block beyond this sample's length (packed batches)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
    const int key0w = kb * KB + wv * KW;  // the wave's first key; group j holds keys key0w + 16 j + [0, 16)

    v8s kf[G][KS], vf[G][KS];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int key = key0w + 16 * j + c;
        load_row_frags<KS>(kf[j], a.k + row_base(a.sk, si, b, h) + (long)key * a.sk.t, key < T, g, a.hd);
        load_row_frags<KS>(vf[j], a.v + row_base(a.sv, si, b, h) + (long)key * a.sv.t, key < T, g, a.hd);
    }
    const uint16_t* qp = a.q + row_base(a.sq, si, b, h);
    const uint16_t* dop = a.dout + row_base(a.sdo, si, b, h);
    const float* lsep = a.lse + ((long)b * a.H + h) * a.T;
    const float* dsp = a.dsum + ((long)b * a.H + h) * a.T;

    v4f dk[G][NT], dv[G][NT];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            dk[j][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
            dv[j][nt] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    const int nqt = (T + kTile - 1) / kTile;
    const int qt0 = CAUSAL ? kb * G : 0;  // the first query tile that sees a key of this workgroup
    TileRegs<HDP> rq, rdo;
    float rstat = 0.0f;  // threads 0..63: lse of query tid (log2 units, +inf when invalid); 64..127: D
    auto load_stats = [&](int qbase) {
        if (tid < 128) {
            const int qq = qbase + (tid & 63);
            if (tid < 64) rstat = qq < T ? lsep[qq] * 1.4426950408889634f : INFINITY;
            else rstat = qq < T ? dsp[qq] : 0.0f;
        }
    };
    const __amdgpu_buffer_rsrc_t qrs = slice_rsrc(qp, a.sq.t, T, a.hd), dors = slice_rsrc(dop, a.sdo.t, T, a.hd);
    const int qst = (int)a.sq.t * 2, dost = (int)a.sdo.t * 2;
    rq.load(qrs, qst, qt0 * kTile);
    rdo.load(dors, dost, qt0 * kTile);
    load_stats(qt0 * kTile);
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        rq.store(sQ, SQ);
        rdo.store(sDO, SQ);
        if (tid < 64) sL[tid] = rstat;
        else if (tid < 128) sD[tid - 64] = rstat;
        __syncthreads();
        if (qt + 1 < nqt) {
            rq.load(qrs, qst, (qt + 1) * kTile);
            rdo.load(dors, dost, (qt + 1) * kTile);
            load_stats((qt + 1) * kTile);
        }
        const int qb0 = qt * kTile;
        if (CAUSAL && qb0 + kTile - 1 < key0w) continue;  // every query of the tile precedes this wave's keys
        const bool need_mask = CAUSAL && qb0 < key0w + KW - 1;  // wave-uniform: some (query, key) pair of this tile is hidden
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (CAUSAL && qb0 + half * 32 + 31 < key0w) continue;
            v4f p[G][2], ds[G][2];
#pragma unroll
            for (int rl = 0; rl < 2; ++rl) {
                const int rt = 2 * half + rl;
                v8s aq[KS], ado[KS];
                load_tile_frags<KS>(aq, sQ, SQ, rt, c, g);
                load_tile_frags<KS>(ado, sDO, SQ, rt, c, g);
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const v4f s = frag_dot<KS>(aq, kf[j]);
                    const v4f dp = frag_dot<KS>(ado, vf[j]);
                    const int key = key0w + 16 * j + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ql = rt * 16 + g * 4 + r;
                        float pe = fast_exp2(__builtin_fmaf(s[r], a.scale_log2, -sL[ql]));  // invalid query: lse = +inf -> 0
                        if (CAUSAL && need_mask && key > qb0 + ql) pe = 0.0f;
                        p[j][rl][r] = pe;
                        ds[j][rl][r] = pe * (dp[r] - sD[ql]);
                    }
                }
            }
            const int ro = (half * 32 + 4 * g + (c >> 2)) * SQ + (c & 3) * 4;
            {
                v8s fr[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fr[nt] = cat8(lds_tr16(&sDO[ro + nt * 16]), lds_tr16(&sDO[ro + 16 * SQ + nt * 16]));
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const v8s pf = pack8(p[j][0], p[j][1]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) dv[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[nt], pf, dv[j][nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) fr[nt] = cat8(lds_tr16(&sQ[ro + nt * 16]), lds_tr16(&sQ[ro + 16 * SQ + nt * 16]));
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const v8s dsf = pack8(ds[j][0], ds[j][1]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) dk[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[nt], dsf, dk[j][nt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int key = key0w + 16 * j + c;
        if (key < T) {
            const long ro = (si.tok0 + key) * (a.hd >> 1);
            store_grad_row<NT>(a.dk + row_base(a.sdk, si, b, h) + (long)key * a.sdk.t, dk[j], a.scale, g, a.hd,
                               a.rope_cos ? a.rope_cos + ro : nullptr, a.rope_cos ? a.rope_sin + ro : nullptr);
            store_grad_row<NT>(a.dv + row_base(a.sdv, si, b, h) + (long)key * a.sdv.t, dv[j], 1.0f, g, a.hd, nullptr, nullptr);
        }
    }
}

// Row groups per wave of each kernel. Measured alone at the bs=64 shapes (tools/attn_sweep.sh, profiles/r05_attn_sweep.txt): two groups
// help the 72-wide SigLIP heads (forward 66.7 -> 61.0, dq 90.5 -> 82.0, dk/dv 115.4 -> 97.4 us), change nothing at 64 (DINOv2) and cost
// occupancy at 128 (Llama: dk/dv 360 -> 423 us) — these kernels are bound by the bytes a CU keeps in flight, not by LDS or MFMA issue
// (MFMA 14 %, LDS 11-29 % busy, 41 % of the wave time in memory waits at 2-3 workgroups per CU). VAA_ATTN_G = "fqk" digits (forward, dq,
// dk/dv; each 1 or 2) overrides the choice for experiments.
static void attn_groups(int hd, int& gf, int& gq, int& gk) {
    static int cfg[3] = {0, 0, 0};
    static std::once_flag once;
    std::call_once(once, [] {
        const char* ev = getenv("VAA_ATTN_G");
        for (int i = 0; i < 3; ++i) cfg[i] = (ev && ev[0] && ev[1] && ev[2] && (ev[i] == '1' || ev[i] == '2')) ? ev[i] - '0' : 0;
    });
    const int dflt = (hd > 64 && hd <= 80) ? 2 : 1;
    gf = cfg[0] ? cfg[0] : dflt;
    gq = cfg[1] ? cfg[1] : dflt;
    gk = cfg[2] ? cfg[2] : dflt;
}

template <bool CAUSAL>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
    int gf, gq, gk;
    attn_groups(a.hd, gf, gq, gk);
    const unsigned pairs8 = (unsigned)(((long)a.B * a.H + 7) / 8 * 8);
    const unsigned grid_q = pairs8 * (unsigned)((a.T + kTile * gq - 1) / (kTile * gq)), grid_k = pairs8 * (unsigned)((a.T + kTile * gk - 1) / (kTile * gk));
#define VAA_ATT_BWD(KS, NT)                                                                                                  \
    do {                                                                                                                     \
        if (gq == 2) hipLaunchKernelGGL((attn_bwd_dq_kernel<KS, NT, CAUSAL, 2>), dim3(grid_q), dim3(256), 0, st, a);         \
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<KS, NT, CAUSAL, 1>), dim3(grid_q), dim3(256), 0, st, a);                 \
        int rc = check_launch("vaa_model_attention_bwd(dq)");                                                                \
        if (rc != VAA_OK) return rc;                                                                                         \
        if (gk == 2) hipLaunchKernelGGL((attn_bwd_dkv_kernel<KS, NT, CAUSAL, 2>), dim3(grid_k), dim3(256), 0, st, a);        \
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<KS, NT, CAUSAL, 1>), dim3(grid_k), dim3(256), 0, st, a);                \
        return check_launch("vaa_model_attention_bwd(dkv)");                                                                 \
    } while (0)
    if (a.hd <= 64) VAA_ATT_BWD(2, 4);
    else if (a.hd <= 80) VAA_ATT_BWD(3, 5);
    else if (a.hd <= 96) VAA_ATT_BWD(3, 6);
    else VAA_ATT_BWD(4, 8);
#undef VAA_ATT_BWD
}

template <bool CAUSAL>
static int launch_fwd(const AttnFwdArgs& a, hipStream_t st) {
    int gf, gq, gk;
    attn_groups(a.hd, gf, gq, gk);
    const unsigned grid = (unsigned)(((long)a.B * a.H + 7) / 8 * 8) * (unsigned)((a.T + kTile * gf - 1) / (kTile * gf));
#define VAA_ATT_FWD(KS, NT)                                                                                          \
    do {                                                                                                             \
        if (gf == 2) hipLaunchKernelGGL((attn_fwd_kernel<KS, NT, CAUSAL, 2>), dim3(grid), dim3(256), 0, st, a);      \
        else hipLaunchKernelGGL((attn_fwd_kernel<KS, NT, CAUSAL, 1>), dim3(grid), dim3(256), 0, st, a);              \
    } while (0)
    if (a.hd <= 64) VAA_ATT_FWD(2, 4);
    else if (a.hd <= 80) VAA_ATT_FWD(3, 5);
    else if (a.hd <= 96) VAA_ATT_FWD(3, 6);
    else VAA_ATT_FWD(4, 8);
#undef VAA_ATT_FWD
    return check_launch("vaa_model_attention_fwd");
}

static bool strides_ok(const int64_t* s) { return s && (s[0] % 8) == 0 && (s[1] % 8) == 0 && (s[2] % 8) == 0; }
static AttnStr mk(const int64_t* s) { AttnStr r; r.b = s[0]; r.t = s[1]; r.h = s[2]; return r; }

}  // namespace vaa

extern "C" int vaa_model_attention_fwd(const uint16_t* q, const int64_t* q_str, const uint16_t* k, const int64_t* k_str, const uint16_t* v,
                                       const int64_t* v_str, uint16_t* o, const int64_t* o_str, float* lse, const int32_t* cu_seqlens, int B,
                                       int H, int T, int hd, int causal, float scale, void* stream) {
    using namespace vaa;
    if (!q || !k || !v || !o || !lse) {
        set_error("vaa_model_attention_fwd: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B <= 0 || H <= 0 || T <= 0 || hd <= 0 || hd > 128 || (hd % 8) != 0 || !strides_ok(q_str) || !strides_ok(k_str) || !strides_ok(v_str) ||
        !strides_ok(o_str)) {
        set_error("vaa_model_attention_fwd: unsupported shape (B=%d H=%d T=%d hd=%d; hd %% 8 == 0, hd <= 128, strides %% 8 == 0)", B, H, T, hd);
        return VAA_E_UNSUPPORTED;
    }
    AttnFwdArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
    a.sq = mk(q_str); a.sk = mk(k_str); a.sv = mk(v_str); a.so = mk(o_str);
    a.cu = cu_seqlens;
    a.B = B; a.H = H; a.T = T; a.hd = hd;
    a.scale_log2 = scale * 1.4426950408889634f;
    return causal ? launch_fwd<true>(a, (hipStream_t)stream) : launch_fwd<false>(a, (hipStream_t)stream);
}

extern "C" int vaa_model_attention_bwd(const uint16_t* q, const int64_t* q_str, const uint16_t* k, const int64_t* k_str, const uint16_t* v,
                                       const int64_t* v_str, const uint16_t* o, const int64_t* o_str, const uint16_t* dout,
                                       const int64_t* do_str, const float* lse, float* dsum, uint16_t* dq, const int64_t* dq_str, uint16_t* dk,
                                       const int64_t* dk_str, uint16_t* dv, const int64_t* dv_str, const float* rope_cos, const float* rope_sin,
                                       const int32_t* cu_seqlens, int B, int H, int T, int hd, int causal, float scale, void* stream) {
    using namespace vaa;
    if (!q || !k || !v || !o || !dout || !lse || !dsum || !dq || !dk || !dv) {
        set_error("vaa_model_attention_bwd: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B <= 0 || H <= 0 || T <= 0 || hd <= 0 || hd > 128 || (hd % 8) != 0 || !strides_ok(q_str) || !strides_ok(k_str) || !strides_ok(v_str) ||
        !strides_ok(o_str) || !strides_ok(do_str) || !strides_ok(dq_str) || !strides_ok(dk_str) || !strides_ok(dv_str)) {
        set_error("vaa_model_attention_bwd: unsupported shape (B=%d H=%d T=%d hd=%d; hd %% 8 == 0, hd <= 128, strides %% 8 == 0)", B, H, T, hd);
        return VAA_E_UNSUPPORTED;
    }
    AttnBwdArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse = lse; a.dsum = dsum; a.dq = dq; a.dk = dk; a.dv = dv;
    a.sq = mk(q_str); a.sk = mk(k_str); a.sv = mk(v_str); a.so = mk(o_str); a.sdo = mk(do_str);
    a.sdq = mk(dq_str); a.sdk = mk(dk_str); a.sdv = mk(dv_str);
    if ((rope_cos != nullptr) != (rope_sin != nullptr) || (rope_cos && hd != 64 && hd != 128)) {
        set_error("vaa_model_attention_bwd: fused rotary adjoint needs both tables and hd in {64, 128} (hd=%d)", hd);
        return VAA_E_UNSUPPORTED;
    }
    a.rope_cos = rope_cos; a.rope_sin = rope_sin;
    a.cu = cu_seqlens;
    a.B = B; a.H = H; a.T = T; a.hd = hd;
    a.scale = scale;
    a.scale_log2 = scale * 1.4426950408889634f;
    return causal ? launch_bwd<true>(a, (hipStream_t)stream) : launch_bwd<false>(a, (hipStream_t)stream);
}
