// vaa_patch_fwd.hip — K1: fused paste + affine warp + mask + dual normalise + bf16 cast, one launch per batch.
//
// Replaces the per-image PyTorch op chain of RandomPatchTransform.apply_random_patch_batch
// (appply_random_transform.py:104-136: ToTensor, -100 canvas, slice paste, affine_grid, grid_sample,
// torch.where, 2x normalise, cat) and the caller's `.to(torch.bfloat16)` (UADA.py:142).
//
// HBM-bound streaming kernel: per image it must read 150,528 B of u8 pixels and write 602,112 B of bf16
// planes. Layout/mapping:
//   * a thread owns 8 consecutive pixels of one row: 24 contiguous input bytes (3 x 8-byte loads) and one
//     16-byte store into each of the 6 output planes; a wave therefore reads 1.5 KiB and writes 6 x 1 KiB
//     contiguous segments.
//   * camera pixels (95 % of the frame) only depend on (u8 value, channel): a 3x256 LUT of packed
//     {bf16 norm0, bf16 norm1} built once per workgroup in LDS with the reference's exact arithmetic
//     (v/255, (x-mean)/std with true divisions, RNE cast) turns them into 3 LDS reads per pixel.
//   * the warp is evaluated only for 8-pixel groups whose source segment can touch the pasted rectangle
//     (both segment end points are real pixels, coordinates are monotone along the segment).
#include "vaa_common.h"

namespace vaa {

struct FwdArgs {
    const uint8_t* img;
    const float* patch;
    const int32_t* xy;
    const float* theta;
    uint16_t* out;
    uint8_t* keep;
    const int32_t* pdesc;  // nullptr: one patch [3,ph,pw] for the batch; else [B,4] = {h_b, w_b, offset_b, 0}: image b pastes patch + offset_b
    int B, ph, pw, geometry, mask_mode;
    Norm6 nrm;
};

__device__ __forceinline__ void patch_of(const FwdArgs& a, int b, int& ph, int& pw, const float*& p) {
    if (a.pdesc) {  // resize_patch=True (appply_random_transform.py:113-118): every image has its own resized patch
        ph = a.pdesc[4 * b]; pw = a.pdesc[4 * b + 1]; p = a.patch + a.pdesc[4 * b + 2];
    } else {
        ph = a.ph; pw = a.pw; p = a.patch;
    }
}

constexpr int kPix = 16;                                  // pixels per item: 48 B in (3 x 16 B), 2 x 16 B out per plane
constexpr int kItemsPerRow = VAA_IMG / kPix;              // 14
constexpr int kItemsPerImg = VAA_IMG * kItemsPerRow;      // 3136
constexpr int kFwdThreads = 256;
constexpr int kMaxRows = 20;                              // rows one workgroup's 256 items can touch (256/14 + 2)

// Host and device share the arithmetic of the camera-pixel LUT: IEEE fp32 true divisions, RNE bf16 cast
// (host code of this file is compiled with -ffp-contract=off as well).
__host__ __device__ inline uint32_t bf16_rne(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__host__ __device__ inline uint32_t norm_pack(float v, const Norm6& n, int c) {
    float o0 = (v - n.mean[c]) / n.stdv[c];
    float o1 = (v - n.mean[c + 3]) / n.stdv[c + 3];
    return bf16_rne(o0) | (bf16_rne(o1) << 16);
}

// Conservative COLUMN span of output row i of image b: pixels [jlo, jhi] may have a source point that touches the patch; jhi < jlo
// (normalised to [0, -1]) when the row is clear. Every role of both K1 kernels derives its ownership from this one function.
__device__ __forceinline__ void row_span(const FwdArgs a, int b, int i, int& jlo, int& jhi) {
    const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
    int ph, pw;
    const float* unused;
    patch_of(a, b, ph, pw, unused);
    jlo = 0; jhi = -1;
    if (a.geometry) {
        float th[6];
#pragma unroll
        for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
        const PixAffine pa = pix_affine(th);
        const float xlo = (px == 0) ? -1e30f : (float)(px - 1);
        const float xhi = (px + pw == VAA_IMG) ? 1e30f : (float)(px + pw);
        const float ylo = (py == 0) ? -1e30f : (float)(py - 1);
        const float yhi = (py + ph == VAA_IMG) ? 1e30f : (float)(py + ph);
        float jl = -1e30f, jh = 1e30f;
        solve_interval(pa.a00, pa.a01 * (float)i + pa.c0, xlo, xhi, jl, jh);
        solve_interval(pa.a10, pa.a11 * (float)i + pa.c1, ylo, yhi, jl, jh);
        if (jl <= jh) {
            jlo = (int)fmaxf(0.0f, floorf(jl) - 1.0f);
            jhi = (int)fminf((float)(VAA_IMG - 1), ceilf(jh) + 1.0f);
        }
    } else if (i >= py && i < py + ph) {
        jlo = px;
        jhi = px + pw - 1;
    }
    // an interval that lies wholly beyond the frame (possible when a side is open to +-infinity: patch on the frame edge) clamps to
    // jhi < jlo with jlo > 0; normalise every empty row to [0, -1] so that ALL roles see "nothing" (the background role of the planar
    // kernel packs the count into 8 bits and would otherwise skip items nobody writes)
    if (jhi < jlo) { jlo = 0; jhi = -1; }
}

// Conservative ITEM bounds of output row i of image b: items [it_lo, it_hi] (16-pixel groups) may contain a pixel whose
// source point touches the patch; it_hi < it_lo when the row is clear. Both roles of the planar kernel call this same function, which
// is what makes their ownership of items disjoint and complete.
__device__ __forceinline__ void row_items(const FwdArgs a, int b, int i, int& it_lo, int& it_hi) {
    int jlo, jhi;
    row_span(a, b, i, jlo, jhi);
    it_lo = jlo >> 4;
    it_hi = (jhi < jlo) ? -1 : (jhi >> 4);
}

// One launch, two workgroup roles:
//   blockIdx.x >= n_fp : BACKGROUND — 256 items of 16 pixels, pure streaming through the LUT; items that may show the
//                        patch are skipped entirely (not written).
//   blockIdx.x <  n_fp : FOOTPRINT  — owns exactly the skipped items of one image (a 1/fsplit share of them), one LANE per
//                        pixel: LUT value, exact warp sample, mask, normalise, 2-byte stores; keep bits by wave ballot.
// The two roles never write the same byte, so no ordering between workgroups is needed.
#ifdef VAA_K1_TIMING
__device__ long long* vaa_k1_dbg = nullptr;
#define K1_T0() long long k1_t = wall_clock64(); const long long k1_start = k1_t; long long k1_acc[6] = {0, 0, 0, 0, 0, 0};
#define K1_STAMP(i) { const long long tn = wall_clock64(); k1_acc[i] += tn - k1_t; k1_t = tn; }
#define K1_FLUSH(role) if ((threadIdx.x & 63) == 0 && vaa_k1_dbg) { long long* d = vaa_k1_dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8; \
    for (int z = 0; z < 6; ++z) d[z] = k1_acc[z]; d[6] = role; d[7] = k1_start; }
#else
#define K1_T0()
#define K1_STAMP(i)
#define K1_FLUSH(role)
#endif

__global__ __launch_bounds__(kFwdThreads) void patch_apply_fwd_kernel(const FwdArgs a, int n_fp, int fsplit) {
    __shared__ uint32_t lut[3 * 256];
    __shared__ float bgrid[VAA_IMG];
    __shared__ uint32_t row_word[VAA_IMG];  // footprint role: (it_lo << 8) | n_items per row; background: first kMaxRows rows
    __shared__ int red_min, red_max, red_n;
    const int tid = threadIdx.x;
    K1_T0()
    const bool bg = (int)blockIdx.x >= n_fp;
    // Background role: the input bytes of this lane's two half items depend on nothing but the item index, so their loads are
    // issued first and fly while the LUT is built and the placement (xy, theta -> row table) is fetched.
    const long item0 = bg ? (long)((int)blockIdx.x - n_fp) * kFwdThreads : 0;
    const long total = (long)a.B * kItemsPerImg;
    uint2 pre[2][3];
    if (bg) {
        const long witem0 = item0 + (long)(tid >> 6) * 64;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const long item = witem0 + seg * 32 + ((tid & 63) >> 1);
            const uint2* src = reinterpret_cast<const uint2*>(a.img + ((size_t)(item < total ? item : 0) * kPix + (tid & 1) * 8) * 3);
            pre[seg][0] = src[0]; pre[seg][1] = src[1]; pre[seg][2] = src[2];
        }
    }
    // camera-pixel LUT: same IEEE arithmetic as the host (true divisions, RNE), 3 entries per thread
#pragma unroll
    for (int e = tid; e < 768; e += kFwdThreads) lut[e] = norm_pack((float)(e & 255) / 255.0f, a.nrm, e >> 8);
    // (building this table once on the host and carrying it in the kernel-argument segment was measured: 16.4 vs 15.5 us at bs=64 —
    // bulk reads from the kernarg segment are slower than nine correctly-rounded divisions per thread)

    if (bg) {  // footprint workgroups come FIRST in dispatch order: their latency chain overlaps the stream
        // ------------------------------------------------------------------ background role
        const long grow0 = item0 / kItemsPerRow;  // first global row (b*224 + i) this workgroup touches
        if (tid < kMaxRows) {
            const long gr = grow0 + tid;
            int lo = 0, hi = -1;
            if (gr < (long)a.B * VAA_IMG) row_items(a, (int)(gr / VAA_IMG), (int)(gr % VAA_IMG), lo, hi);
            row_word[tid] = ((uint32_t)(lo & 0xff) << 8) | (uint32_t)((hi - lo + 1) & 0xff);
        }
        K1_STAMP(0)
        __syncthreads();
        K1_STAMP(1)
        // A wave owns 64 consecutive items = 1024 pixels. Lane l handles two HALF items: pixels [8l, 8l+8) of the wave's first
        // 512 pixels and the same of its second 512, so that every store instruction writes 64 x 16 B = 1 KB CONTIGUOUS bytes
        // of a plane (a thread storing its own 16 pixels as two 16 B halves leaves every 128 B line to be completed by a
        // second instruction, which halves the write rate). Input: 24 B per half item as three 8 B loads at lane stride 24 B.
        const int wv = tid >> 6, lane = tid & 63;
        const long witem0 = item0 + (long)wv * 64;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const long item = witem0 + seg * 32 + (lane >> 1);
            if (item >= total) continue;
            const long grow = item / kItemsPerRow;
            const int it = (int)(item - grow * kItemsPerRow);
            const uint32_t w = row_word[(int)(grow - grow0)];
            const int lo = (int)(w >> 8), n = (int)(w & 0xffu);
            if (it >= lo && it < lo + n) continue;  // owned by the footprint role
            const int half = lane & 1;
            const int j0 = it * kPix + half * 8;
            const int b = (int)(grow / VAA_IMG), i = (int)(grow - (long)b * VAA_IMG);
            const uint32_t d[6] = {pre[seg][0].x, pre[seg][0].y, pre[seg][1].x, pre[seg][1].y, pre[seg][2].x, pre[seg][2].y};
            uint32_t L[3][8];  // packed {bf16 plane c, bf16 plane c+3} per pixel
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int k = p * 3 + c;
                    L[c][p] = lut[c * 256 + ((d[k >> 2] >> (8 * (k & 3))) & 0xffu)];
                }
            const size_t obase = ((size_t)b * 6 * VAA_IMG + i) * VAA_IMG + j0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                uint32_t lo4[4], hi4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    lo4[q] = __builtin_amdgcn_perm(L[c][2 * q + 1], L[c][2 * q], 0x05040100u);
                    hi4[q] = __builtin_amdgcn_perm(L[c][2 * q + 1], L[c][2 * q], 0x07060302u);
                }
                *reinterpret_cast<uint4*>(a.out + obase + (size_t)c * VAA_NPIX) = make_uint4(lo4[0], lo4[1], lo4[2], lo4[3]);
                *reinterpret_cast<uint4*>(a.out + obase + (size_t)(c + 3) * VAA_NPIX) = make_uint4(hi4[0], hi4[1], hi4[2], hi4[3]);
            }
            if (a.keep) {  // one byte of the keep mask per half item
                const size_t kbase = ((size_t)b * 3 * VAA_NPIX + (size_t)i * VAA_IMG + j0) >> 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) a.keep[kbase + (size_t)c * (VAA_NPIX / 8)] = (uint8_t)0;
            }
        }
        K1_STAMP(2)
        K1_FLUSH(0)
        return;
    }

    // ---------------------------------------------------------------------- footprint role
    const int fid = (int)blockIdx.x;
    const int b = fid / fsplit, chunk = fid - b * fsplit;
    if (tid == 0) { red_min = VAA_IMG; red_max = -1; red_n = 0; }
    if (tid < VAA_IMG) bgrid[tid] = base_coord(tid);
    __syncthreads();
    {   // row table + footprint extent; same-address LDS atomics serialise, so reduce in the wave first (4 atomics, not 224)
        int lo = 0, hi = -1;
        if (tid < VAA_IMG) row_items(a, b, tid, lo, hi);
        const int n = hi - lo + 1;
        if (tid < VAA_IMG) row_word[tid] = ((uint32_t)(lo & 0xff) << 8) | (uint32_t)(n > 0 ? n : 0);
        const int wmin = wave_min_i(n > 0 ? tid : VAA_IMG), wmax = wave_max_i(n > 0 ? tid : -1), wn = wave_max_i(n > 0 ? n : 0);
        if ((tid & 63) == 0) { atomicMin(&red_min, wmin); atomicMax(&red_max, wmax); atomicMax(&red_n, wn); }
    }
    __syncthreads();
    K1_STAMP(0)
    const int rmin = red_min, nrows = red_max - rmin + 1;
    if (nrows <= 0) return;
    const int nseg = (red_n + 1) >> 1;                             // half-wave = 32 pixels = 2 items per slot
    const uint32_t inv_nseg = (65536u + nseg - 1) / nseg;
    const int nslots = nrows * nseg;
    const int hw = (chunk * kFwdThreads + tid) >> 5, nhw = fsplit * (kFwdThreads >> 5), hlane = tid & 31;
    const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
    int ph, pw;
    const float* patch;
    patch_of(a, b, ph, pw, patch);
    const int plane = ph * pw;
    float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (a.geometry) {
#pragma unroll
        for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
    }
    for (int sidx = hw; sidx < nslots; sidx += nhw) {  // trip count is uniform within a wave (both half-waves iterate alike or idle)
        const int r = (int)(((uint32_t)sidx * inv_nseg) >> 16);
        const int k = sidx - r * nseg;
        const int i = rmin + r;
        const uint32_t w = row_word[i];
        const int off = (k << 5) + hlane;
        const bool active = (off >> 4) < (int)(w & 0xffu);
        const int j = ((int)(w >> 8) << 4) + off;
        uint32_t L[3] = {0u, 0u, 0u};
        bool kept[3] = {false, false, false};
        if (active) {
            // every load of this pixel (3 camera bytes, 12 patch texels) is issued before anything is consumed: texel addresses
            // are clamped into the patch so the gather is branch-free, the canvas rule (-100 outside the paste, 0 beyond the
            // frame) is applied to the loaded values afterwards -> one memory latency per slot instead of two.
            const uint8_t* sp = a.img + ((size_t)(b * VAA_IMG + i) * VAA_IMG + j) * 3;
            const uint32_t by0 = sp[0], by1 = sp[1], by2 = sp[2];
            float cv[3];
            bool inside;
            if (a.geometry) {
                const Samp s = sample_pos(bgrid[j], bgrid[i], th);
                const int u0 = s.x0 - px, v0 = s.y0 - py;
                inside = !(u0 < -1 || u0 >= pw || v0 < -1 || v0 >= ph);
                const int uc0 = min(max(u0, 0), pw - 1), uc1 = min(max(u0 + 1, 0), pw - 1);
                const int vc0 = min(max(v0, 0), ph - 1), vc1 = min(max(v0 + 1, 0), ph - 1);
                float t[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* pc = patch + c * plane;
                    t[c][0] = pc[vc0 * pw + uc0]; t[c][1] = pc[vc0 * pw + uc1];
                    t[c][2] = pc[vc1 * pw + uc0]; t[c][3] = pc[vc1 * pw + uc1];
                }
                const bool ux0 = (unsigned)u0 < (unsigned)pw, ux1 = (unsigned)(u0 + 1) < (unsigned)pw;
                const bool vy0 = (unsigned)v0 < (unsigned)ph, vy1 = (unsigned)(v0 + 1) < (unsigned)ph;
                const bool fx1 = s.x0 + 1 < VAA_IMG, fy1 = s.y0 + 1 < VAA_IMG;  // (x0, y0) itself is always inside the frame
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float vnw = (ux0 && vy0) ? t[c][0] : -100.0f;
                    const float vne = !fx1 ? 0.0f : ((ux1 && vy0) ? t[c][1] : -100.0f);
                    const float vsw = !fy1 ? 0.0f : ((ux0 && vy1) ? t[c][2] : -100.0f);
                    const float vse = !(fx1 && fy1) ? 0.0f : ((ux1 && vy1) ? t[c][3] : -100.0f);
                    cv[c] = __builtin_fmaf(vse, s.se, __builtin_fmaf(vsw, s.sw, __builtin_fmaf(vne, s.ne, vnw * s.nw)));
                }
            } else {
                const int u = j - px, v = i - py;
                inside = (unsigned)u < (unsigned)pw && (unsigned)v < (unsigned)ph;
                const int uc = min(max(u, 0), pw - 1), vc = min(max(v, 0), ph - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) cv[c] = patch[c * plane + vc * pw + uc];
            }
            K1_STAMP(1)
            L[0] = lut[by0]; L[1] = lut[256 + by1]; L[2] = lut[512 + by2];
            K1_STAMP(2)
            if (inside) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (keep_rule(cv[c], a.mask_mode)) { L[c] = norm_pack(cv[c], a.nrm, c); kept[c] = true; }
            }
            const size_t o = ((size_t)b * 6 * VAA_IMG + i) * VAA_IMG + j;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.out[o + (size_t)c * VAA_NPIX] = (uint16_t)(L[c] & 0xffffu);
                a.out[o + (size_t)(c + 3) * VAA_NPIX] = (uint16_t)(L[c] >> 16);
            }
        }
        if (a.keep) {  // 16 consecutive lanes = one item = one 16-bit word of the keep mask
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned long long m = __ballot(kept[c]);
                if (active && (hlane & 15) == 0) {
                    const uint32_t bits = (uint32_t)(m >> (tid & 48)) & 0xffffu;
                    const size_t kbase = ((size_t)b * 3 * VAA_NPIX + (size_t)i * VAA_IMG + j) >> 3;
                    *reinterpret_cast<uint16_t*>(a.keep + kbase + (size_t)c * (VAA_NPIX / 8)) = (uint16_t)bits;
                }
            }
        }
        K1_STAMP(3)
    }
    K1_FLUSH(1)
}


// ---------------------------------------------------------------------------------------------------------------------------------
// K1 in TILE-MAJOR form (vaa_patch_apply_fwd_tiles): the same values, written as the two ViT patch-embed GEMM operands
//   out0 / out1 [B,256,588] bf16 : tile t = ty*16 + tx of the 14x14 tiling, element c*196 + y*14 + x (timm PatchEmbed's im2col order,
//                                  modeling_prismatic.py:120-123) — what `unfold(pixel_values)` would produce, without the two 19 MB
//                                  permute copies per step;
//   keep_t [B,3,256,14] u16      : bit x of word (c, t, y) = channel c of pixel (14 ty + y, 14 tx + x) shows the patch;
//   flags  [B,256] u32           : != 0 when tile t holds a kept pixel (any channel): the tile list of K2' without its 42-load flag pass.
// Ownership is per TILE: a tile is a footprint tile when the conservative column span (row_span) of any of its 14 rows meets it.
//   BACKGROUND workgroup = one tile row (b, ty): 9,408 contiguous input bytes staged through LDS, every thread converts pixel PAIRS
//       (6 input bytes -> one dword per channel and tower) so that consecutive lanes write consecutive dwords of a tile; footprint
//       tiles are skipped; keep words / flags of its background tiles are cleared.
//   FOOTPRINT workgroup (dispatched first) = a share of the footprint tiles of one image, one tile per WAVE, one lane per pixel in four
//       passes of 56 lanes whose loads are all in flight together: exact warp sample, mask, normalise; keep words by ballot, the tile flag
//       by the wave.
struct TileOut {
    uint16_t* out0;
    uint16_t* out1;
    uint16_t* keep_t;
    uint32_t* flags;  // byte w of word t: wave w of the footprint workgroup saw a kept pixel in its rows of tile t
};

constexpr int kTS = 14, kTPS = 16, kTElems = 3 * kTS * kTS;  // tile side, tiles per side, elements per tile and tower (588)
constexpr int kRowBytes = VAA_IMG * 3;                         // 672
constexpr int kTileRowBytes = kTS * kRowBytes;                 // 9,408 = 588 x 16 B
constexpr int kPairSlots = kTPS * kTS * (kTS / 2);             // 1,568 pixel pairs per tile row

__device__ __forceinline__ bool tile_meets(const int* lo, const int* hi, int tx) {  // lo/hi: the 14 row spans of the tile row
    bool f = false;
#pragma unroll
    for (int y = 0; y < kTS; ++y) f = f || (hi[y] >= lo[y] && lo[y] <= kTS * tx + kTS - 1 && hi[y] >= kTS * tx);
    return f;
}

__global__ __launch_bounds__(kFwdThreads) void patch_apply_tiles_kernel(const FwdArgs a, const TileOut o, int n_fp, int fsplit) {
    __shared__ uint32_t lut[3 * 256];
    __shared__ __align__(16) uint8_t inbuf[kTileRowBytes];  // background: the tile row's input bytes; footprint: bgrid + tables (aliased below)
    __shared__ int sp_lo[VAA_IMG], sp_hi[VAA_IMG];
    __shared__ int16_t tiles[256];
    __shared__ int wave_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool bg = (int)blockIdx.x >= n_fp;
    uint4 pre[3];
    int b, ty = 0;
    if (bg) {
        const int id = (int)blockIdx.x - n_fp;
        b = id >> 4; ty = id & 15;
        const uint4* src = reinterpret_cast<const uint4*>(a.img + ((size_t)b * VAA_IMG + kTS * ty) * kRowBytes);  // 16-byte aligned: 672 = 42 x 16
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + k * kFwdThreads;
            pre[k] = idx < kTileRowBytes / 16 ? src[idx] : make_uint4(0, 0, 0, 0);
        }
    } else {
        b = (int)blockIdx.x / fsplit;
    }
#pragma unroll
    for (int e = tid; e < 768; e += kFwdThreads) lut[e] = norm_pack((float)(e & 255) / 255.0f, a.nrm, e >> 8);

    if (bg) {
        // ------------------------------------------------------------------ background role: tile row (b, ty)
        if (tid < kTS) row_span(a, b, kTS * ty + tid, sp_lo[tid], sp_hi[tid]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = tid + k * kFwdThreads;
            if (idx < kTileRowBytes / 16) reinterpret_cast<uint4*>(inbuf)[idx] = pre[k];
        }
        __syncthreads();
        __shared__ uint8_t fp_tile[kTPS];
        if (tid < kTPS) fp_tile[tid] = tile_meets(sp_lo, sp_hi, tid) ? 1 : 0;
        __syncthreads();
        const size_t tbase = ((size_t)b * 256 + ty * kTPS) * kTElems;
        for (int s = tid; s < kPairSlots; s += kFwdThreads) {
            const int t = s / 98, r = s - t * 98;
            if (fp_tile[t]) continue;  // owned by the footprint role
            const int y = r / 7, xp = r - y * 7;
            const uint8_t* ip = inbuf + (y * VAA_IMG + kTS * t + 2 * xp) * 3;  // 6 bytes, 2-byte aligned
            const uint32_t h0 = *reinterpret_cast<const uint16_t*>(ip), h1 = *reinterpret_cast<const uint16_t*>(ip + 2),
                           h2 = *reinterpret_cast<const uint16_t*>(ip + 4);
            const uint32_t A0 = lut[h0 & 255u], A1 = lut[256 + (h0 >> 8)], A2 = lut[512 + (h1 & 255u)];       // pixel 0: channels 0..2
            const uint32_t B0 = lut[h1 >> 8], B1 = lut[256 + (h2 & 255u)], B2 = lut[512 + (h2 >> 8)];         // pixel 1
            const size_t e = tbase + (size_t)t * kTElems + y * kTS + 2 * xp;                                    // even: dword aligned
            *reinterpret_cast<uint32_t*>(o.out0 + e) = __builtin_amdgcn_perm(B0, A0, 0x05040100u);
            *reinterpret_cast<uint32_t*>(o.out1 + e) = __builtin_amdgcn_perm(B0, A0, 0x07060302u);
            *reinterpret_cast<uint32_t*>(o.out0 + e + 196) = __builtin_amdgcn_perm(B1, A1, 0x05040100u);
            *reinterpret_cast<uint32_t*>(o.out1 + e + 196) = __builtin_amdgcn_perm(B1, A1, 0x07060302u);
            *reinterpret_cast<uint32_t*>(o.out0 + e + 392) = __builtin_amdgcn_perm(B2, A2, 0x05040100u);
            *reinterpret_cast<uint32_t*>(o.out1 + e + 392) = __builtin_amdgcn_perm(B2, A2, 0x07060302u);
        }
        if (o.keep_t) {  // keep words of the background tiles: 3 channels x 16 tiles x 14 rows
            for (int w = tid; w < 3 * kTPS * kTS; w += kFwdThreads) {
                const int c = w / (kTPS * kTS), rem = w - c * (kTPS * kTS), t = rem / kTS;
                if (!fp_tile[t]) o.keep_t[(((size_t)b * 3 + c) * 256 + ty * kTPS) * kTS + rem] = (uint16_t)0;
            }
        }
        if (o.flags && tid < kTPS && !fp_tile[tid]) o.flags[(size_t)b * 256 + ty * kTPS + tid] = 0u;
        return;
    }

    // ---------------------------------------------------------------------- footprint role: footprint tiles of image b
    float* bgrid = reinterpret_cast<float*>(inbuf);  // 224 floats
    const int chunk = (int)blockIdx.x - b * fsplit;
    if (tid < VAA_IMG) { bgrid[tid] = base_coord(tid); row_span(a, b, tid, sp_lo[tid], sp_hi[tid]); }
    __syncthreads();
    const bool flag = tile_meets(sp_lo + kTS * (tid >> 4), sp_hi + kTS * (tid >> 4), tid & 15);  // thread t = tile t
    const unsigned long long mk = __ballot(flag);
    if (lane == 0) wave_cnt[wv] = __popcll(mk);
    __syncthreads();
    int base = 0, M = 0;
    for (int q = 0; q < 4; ++q) { if (q < wv) base += wave_cnt[q]; M += wave_cnt[q]; }
    if (flag) tiles[base + __popcll(mk & ((1ull << lane) - 1ull))] = (int16_t)tid;
    __syncthreads();
    const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
    int ph, pw;
    const float* patch;
    patch_of(a, b, ph, pw, patch);
    const int plane = ph * pw;
    float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (a.geometry) {
#pragma unroll
        for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
    }
    // One tile per iteration, one lane per pixel: wave w takes rows 4w..4w+3 (14 lanes = one row = one keep word). Every load of a pixel
    // (3 camera bytes, 12 patch texels at addresses clamped into the patch) is issued before anything is consumed; the canvas rule (-100
    // outside the paste, 0 beyond the frame) is applied to the loaded values. No barrier in the loop: byte w of a tile's flag word is wave w's.
    const int yr = lane / kTS, x = lane - yr * kTS, y = 4 * wv + yr;
    const bool active = lane < 4 * kTS && y < kTS;
    for (int ti = chunk; ti < M; ti += fsplit) {
        const int t = tiles[ti], tty = t >> 4, ttx = t & 15;
        const int i = kTS * tty + (active ? y : 0), j = kTS * ttx + (active ? x : 0);
        uint32_t L[3] = {0u, 0u, 0u};
        bool kept[3] = {false, false, false};
        if (active) {
            const uint8_t* sp = a.img + ((size_t)(b * VAA_IMG + i) * VAA_IMG + j) * 3;
            const uint32_t by0 = sp[0], by1 = sp[1], by2 = sp[2];
            float cv[3];
            bool inside;
            int rx0 = j, ry0 = i;
            float rwf = 0.0f, rnf = 0.0f;
            if (a.geometry) {
                sample_pos_frac(bgrid[j], bgrid[i], th, rx0, ry0, rwf, rnf);
                const Samp s = samp_from_frac(rx0, ry0, rwf, rnf);  // the products of sample_pos, bit for bit
                const int u0 = s.x0 - px, v0 = s.y0 - py;
                inside = !(u0 < -1 || u0 >= pw || v0 < -1 || v0 >= ph);
                const int uc0 = min(max(u0, 0), pw - 1), uc1 = min(max(u0 + 1, 0), pw - 1);
                const int vc0 = min(max(v0, 0), ph - 1), vc1 = min(max(v0 + 1, 0), ph - 1);
                float tx4[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* pc = patch + c * plane;
                    tx4[c][0] = pc[vc0 * pw + uc0]; tx4[c][1] = pc[vc0 * pw + uc1];
                    tx4[c][2] = pc[vc1 * pw + uc0]; tx4[c][3] = pc[vc1 * pw + uc1];
                }
                const bool ux0 = (unsigned)u0 < (unsigned)pw, ux1 = (unsigned)(u0 + 1) < (unsigned)pw;
                const bool vy0 = (unsigned)v0 < (unsigned)ph, vy1 = (unsigned)(v0 + 1) < (unsigned)ph;
                const bool fx1 = s.x0 + 1 < VAA_IMG, fy1 = s.y0 + 1 < VAA_IMG;  // (x0, y0) itself is always inside the frame
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float vnw = (ux0 && vy0) ? tx4[c][0] : -100.0f;
                    const float vne = !fx1 ? 0.0f : ((ux1 && vy0) ? tx4[c][1] : -100.0f);
                    const float vsw = !fy1 ? 0.0f : ((ux0 && vy1) ? tx4[c][2] : -100.0f);
                    const float vse = !(fx1 && fy1) ? 0.0f : ((ux1 && vy1) ? tx4[c][3] : -100.0f);
                    cv[c] = __builtin_fmaf(vse, s.se, __builtin_fmaf(vsw, s.sw, __builtin_fmaf(vne, s.ne, vnw * s.nw)));
                }
            } else {
                const int u = j - px, v = i - py;
                inside = (unsigned)u < (unsigned)pw && (unsigned)v < (unsigned)ph;
                const int uc = min(max(u, 0), pw - 1), vc = min(max(v, 0), ph - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) cv[c] = patch[c * plane + vc * pw + uc];
            }
            L[0] = lut[by0]; L[1] = lut[256 + by1]; L[2] = lut[512 + by2];
            if (inside) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (keep_rule(cv[c], a.mask_mode)) { L[c] = norm_pack(cv[c], a.nrm, c); kept[c] = true; }
            }
            const size_t e = ((size_t)b * 256 + t) * kTElems + y * kTS + x;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                o.out0[e + c * 196] = (uint16_t)(L[c] & 0xffffu);
                o.out1[e + c * 196] = (uint16_t)(L[c] >> 16);
            }
        }
        unsigned long long many = 0ull;
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // 14 consecutive lanes = one row of the tile = one keep word
            const unsigned long long m = __ballot(kept[c]);
            many |= m;
            if (o.keep_t && active && x == 0)
                o.keep_t[(((size_t)b * 3 + c) * 256 + t) * kTS + y] = (uint16_t)((m >> lane) & 0x3fffull);
        }
        if (o.flags && lane == 0) reinterpret_cast<uint8_t*>(o.flags + (size_t)b * 256 + t)[wv] = (uint8_t)(many != 0ull ? 1 : 0);
    }
}

}  // namespace vaa

namespace vaa {

static int launch_patch_apply(const char* who, const uint8_t* img_u8, const float* patch, const int32_t* pdesc, const int32_t* xy,
                              const float* theta, int B, int ph, int pw, int geometry, int mask_mode, const float* mean6,
                              const float* std6, uint16_t* out_bf16, uint8_t* keep_bits, void* stream) {
    if (B == 0) return VAA_OK;  // empty batch: nothing to read or write (pointers may be null)
    if (!img_u8 || !patch || !xy || !out_bf16 || !mean6 || !std6 || (geometry && !theta)) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0 || (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("%s: bad sizes/mode (B=%d ph=%d pw=%d mask_mode=%d)", who, B, ph, pw, mask_mode);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("%s: patch %dx%d larger than the %dx%d frame", who, ph, pw, VAA_IMG, VAA_IMG);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        // the reference pairs `canvas != -100` only with the un-warped paste (paste_patch_fix / random_paste_patch, :138-188); after a
        // warp the all-background blend -100*(nw+ne+sw+se) is not exactly -100, so the rule would depend on rounding over the whole frame
        set_error("%s: VAA_MASK_NE_M100 is defined for geometry=0 only (appply_random_transform.py:153,179)", who);
        return VAA_E_UNSUPPORTED;
    }
    FwdArgs a;
    a.img = img_u8; a.patch = patch; a.xy = xy; a.theta = theta; a.out = out_bf16; a.keep = keep_bits; a.pdesc = pdesc;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) { a.nrm.mean[q] = mean6[q]; a.nrm.stdv[q] = std6[q]; }
    const long total = (long)B * kItemsPerImg;
    const long n_bg = (total + kFwdThreads - 1) / kFwdThreads;
    int fsplit = 16;  // footprint workgroups per image: ~6,400 pixel-lanes per 50x50 footprint -> 2 slot rounds each
    while (fsplit > 1 && (long)B * fsplit > 1024) fsplit >>= 1;  // ~1024 footprint workgroups in total is the measured optimum
    const long n_fp = (long)B * fsplit;
    VAA_LAUNCH(patch_apply_fwd_kernel, dim3((unsigned)(n_fp + n_bg)), dim3(kFwdThreads), 0, (hipStream_t)stream, a,
                       (int)n_fp, fsplit);
    return check_launch(who);
}

}  // namespace vaa

#ifdef VAA_K1_TIMING
extern "C" int vaa_k1_set_debug(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(vaa::vaa_k1_dbg), &p, sizeof(p)); }
#endif

extern "C" int vaa_patch_apply_fwd(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, int B,
                                   int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6,
                                   uint16_t* out_bf16, uint8_t* keep_bits, void* stream) {
    return vaa::launch_patch_apply("vaa_patch_apply_fwd", img_u8, patch, nullptr, xy, theta, B, ph, pw, geometry, mask_mode, mean6, std6,
                                   out_bf16, keep_bits, stream);
}

// K1 with one patch per image (resize_patch=True): image b pastes packed + pdesc[b].offset as [3,h_b,w_b]; max_h/max_w bound the sizes.
extern "C" int vaa_patch_apply_fwd_multi(const uint8_t* img_u8, const float* packed, const int32_t* pdesc, const int32_t* xy,
                                         const float* theta, int B, int max_h, int max_w, int geometry, int mask_mode,
                                         const float* mean6, const float* std6, uint16_t* out_bf16, uint8_t* keep_bits, void* stream) {
    if (B > 0 && !pdesc) {
        vaa::set_error("vaa_patch_apply_fwd_multi: null pdesc");
        return VAA_E_INVALID;
    }
    return vaa::launch_patch_apply("vaa_patch_apply_fwd_multi", img_u8, packed, pdesc, xy, theta, B, max_h, max_w, geometry, mask_mode, mean6,
                                   std6, out_bf16, keep_bits, stream);
}

// K1 in tile-major form (see patch_apply_tiles_kernel): the operands of the two ViT patch-embed GEMMs + tile-major keep words + tile flags.
namespace vaa {
static int patch_apply_tiles_impl(const char* who, const uint8_t* img_u8, const float* patch, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                  int B, int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6, uint16_t* out0,
                                  uint16_t* out1, uint16_t* keep_tiles, uint32_t* tile_flags, void* stream) {
    if (B == 0) return VAA_OK;
    if (!img_u8 || !patch || !xy || !out0 || !out1 || !mean6 || !std6 || (geometry && !theta)) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0 || (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("%s: bad sizes/mode (B=%d ph=%d pw=%d mask_mode=%d)", who, B, ph, pw, mask_mode);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("%s: patch %dx%d larger than the %dx%d frame", who, ph, pw, VAA_IMG, VAA_IMG);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        set_error("%s: VAA_MASK_NE_M100 is defined for geometry=0 only (appply_random_transform.py:153,179)", who);
        return VAA_E_UNSUPPORTED;
    }
    FwdArgs a;
    a.img = img_u8; a.patch = patch; a.xy = xy; a.theta = theta; a.out = nullptr; a.keep = nullptr; a.pdesc = pdesc;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) { a.nrm.mean[q] = mean6[q]; a.nrm.stdv[q] = std6[q]; }
    TileOut o;
    o.out0 = out0; o.out1 = out1; o.keep_t = keep_tiles; o.flags = tile_flags;
    // footprint workgroups per image: each pays a ~2 us prologue (LUT, 224 row spans, tile list) before its first tile, so FEWER, longer-lived
    // workgroups win here — measured at bs=64 with 1024 / 512 / 256 / 128 in total: 18.7 / 16.2 / 18.6 / 25.8 us (a 50x50 footprint meets ~45
    // tiles: ~6 per workgroup at 8 per image); small batches keep 16 per image
    int fsplit = 16;
    while (fsplit > 1 && (long)B * fsplit > 512) fsplit >>= 1;
    const long n_fp = (long)B * fsplit, n_bg = (long)B * 16;
    VAA_LAUNCH(patch_apply_tiles_kernel, dim3((unsigned)(n_fp + n_bg)), dim3(kFwdThreads), 0, (hipStream_t)stream, a, o, (int)n_fp, fsplit);
    return check_launch(who);
}
}  // namespace vaa

extern "C" int vaa_patch_apply_fwd_tiles(const uint8_t* img_u8, const float* patch, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                         int B, int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6,
                                         uint16_t* out0, uint16_t* out1, uint16_t* keep_tiles, uint32_t* tile_flags, void* stream) {
    return vaa::patch_apply_tiles_impl("vaa_patch_apply_fwd_tiles", img_u8, patch, pdesc, xy, theta, B, ph, pw, geometry, mask_mode, mean6, std6, out0, out1,
                                       keep_tiles, tile_flags, stream);
}
