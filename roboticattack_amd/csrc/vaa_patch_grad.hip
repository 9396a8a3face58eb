// vaa_patch_grad.hip — K2: dL/d patch from the model's bf16 pixel gradient, one launch (+ a small reduce) per batch.
//
// Replaces the autograd backward of RandomPatchTransform.apply_random_patch_batch reached from
// `MSE_Distance.backward()` (UADA.py:148): bf16->f32 cast, (g/std0 + g/std1), torch.where mask,
// grid_sample backward (bilinear scatter, padding 'border'), canvas slice, sum over the batch.
//
// Design for gfx950:
//   * only the warped footprint of the patch is touched: per image and output row the kernel bounds the columns whose
//     source point can land on the pasted rectangle (inverse of the pixel-space affine, +-1 px margin); footprint rows are
//     cut into 32-column segments that are dealt round-robin to half-waves, so consecutive lanes read consecutive pixels
//     of a row of each gradient plane. The exact forward coordinates are recomputed per pixel: bounds only need to be
//     conservative.
//   * a workgroup works in ROUNDS of K slots per thread: every load of a round (6 gradient values + 3 keep bytes per pixel)
//     is in flight before the first is consumed (memory-level parallelism instead of occupancy), then the round's largest
//     |G| is reduced over the workgroup.
//   * accumulation is INTEGER: each bilinear contribution fl(G*w) — the very fp32 product grid_sample's backward forms — is rounded
//     to an integer of < 2^30 relative to the workgroup's gradient exponent (one v_cvt_rpi_i32_f32; the power-of-two scale commutes
//     with the product's rounding; re-scaled lazily when a later image has a larger exponent) and added into a patch-shaped int64 tile
//     in LDS with ds_add_u64. Integer sums do not depend on arrival order, so the result is bitwise reproducible for every patch size
//     without serialising the scatter (the reference's CUDA grid_sample backward is not), and ds_add_u64 measured 12.9 cycles per
//     wave-instruction on the footprint access pattern against 20.9 for ds_add_f64 (tools/probe/lds_atomic_probe.hip). The sum is
//     exact to 2^-30 of the largest gradient: against the fp64-accumulated sum of the same products the output differs by its own
//     final fp32 rounding only (7e-8), the reference's fp32 scan-order accumulation by 3e-7 (tools/scratch/k2prec.py).
//   * workgroups are persistent over images and keep accumulating into the same tile (the tile is in patch coordinates),
//     then write their rows of ONE fp32 partial per workgroup-row; a second small kernel adds the partials in fixed order. No global atomics.
//   * ROW BANDS (grid.z): a workgroup owns a band of PATCH rows — it bounds and walks only the part of each footprint whose source points
//     fall on its rows, keeps and drains only those rows. Small batches are spread over the chip this way without extra partial tiles,
//     and patches whose int64 plane exceeds the LDS are covered up to 224x224.
//   * border padding: out-of-frame source points clamp onto frame-edge canvas pixels; when the patch touches the frame edge
//     the row bounds are opened to infinity on that side, everything else is unchanged.
//   * MULTI (resize_patch=True, config 5): one patch PER IMAGE (pdesc), the output is every image's own gradient.
#include <type_traits>

#include "vaa_common.h"

namespace vaa {

struct GradArgs {
    const uint16_t* g;
    const float* patch;
    const int32_t* xy;
    const float* theta;
    const uint8_t* keep;
    float* partial;          // uniform patch: [workgroups][3*ph*pw] partial tiles; MULTI: gpacked (same layout as `patch`)
    const int32_t* pdesc;    // MULTI: [B,4] = {h, w, offset in floats, 0}
    int B, ph, pw, geometry, mask_mode;  // MULTI: ph/pw are upper bounds (LDS sizing)
    int band_rows;           // patch rows per row band (== ph when the whole plane fits the LDS)
    float istd6[6];          // 1/std, rounded from double on the host
    // TILED source (vaa_patch_embed_grad_gather): instead of the 6-plane bf16 pixel gradient `g`, the already combined and
    // scaled gradient of the tiles that carry kept pixels: geff[b][ty*16 + tx][c*196 + y*14 + x] (tiles without a kept pixel are not written)
    const float* geff;
    const float* geff2;      // non-null: the tile kernel ran one tower per workgroup and `geff` holds {tower 0, tower 1} PAIRS per element: the gather adds the two
    int geff_bf16;           // geff / geff2 are the two towers' bf16-ROUNDED pixel gradients, one plane each, unscaled: the gather applies 1/std and adds — the
                             // products and the sum the tile kernel would have formed, bit for bit, at half the bytes written and read back
    const uint16_t* keep_t;  // TILED only: K1's tile-major keep words [B,3,256,14] (vaa_patch_apply_fwd_tiles) instead of `keep`
};

constexpr int kTilePx = 14, kTilesPerSide = 16, kTileElems = 3 * kTilePx * kTilePx;  // ViT patch-embed tiling of the 224x224 frame

#ifdef VAA_K2_TIMING
__device__ long long* vaa_k2_dbg = nullptr;
#endif
constexpr int kImgsPerPass = 8;   // images whose row tables are built together (one barrier set per pass)
constexpr int kCBits = 30;        // a contribution fl(G*w) is rounded to an integer of < 2^30 relative to the workgroup's exponent
constexpr int kExpUnset = -100000;
constexpr long kFlushPixels = 1l << 30;  // footprint pixels a tile may absorb before it is flushed (int64 headroom 2^63 / 2^30 / 4: never in practice)

__device__ __forceinline__ long long shift_round(long long v, int d) {  // v / 2^d, round half up; d in [1, 62]
    return (v + (1ll << (d - 1))) >> d;
}

// floor(x + 0.5) as int32 in ONE instruction (|x| < 2^31); the tie rule differs from round-to-nearest-even only at exact .5, any fixed
// rule keeps the sums order-independent
__device__ __forceinline__ int round_half_up(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Schedule: image b is owned by workgroup-row (b % gx) = blockIdx.x; blockIdx.z selects a band of patch rows (see grad_sched), blockIdx.y
// the channel of the one-channel-per-workgroup instantiations.
//
// Footprint walk without any per-pixel search: the footprint rows [rmin, rmax] of an image are cut into 64-column
// segments starting at an even column; one half-wave owns one (row, segment) slot at a time and a lane owns TWO adjacent
// pixels of it (one 4-byte load per gradient plane, one keep byte), so a lane needs ONE LDS read (packed {jlo,len} of its
// row) to know its pixels. Slots are dealt round-robin to the half-waves of the workgroup-row.
template <int NCH, bool TILED, bool MULTI, bool HASK, int THREADS, int K>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void patch_grad_scatter_kernel(GradArgs a, int gx) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned long long* tile = reinterpret_cast<unsigned long long*>(smem_raw);  // [band_rows*pw][NCH] (channels interleaved)
    __shared__ float bgrid[VAA_IMG];
    __shared__ uint32_t row_word[kImgsPerPass][VAA_IMG];  // (jlo << 16) | len, jlo even
    __shared__ int row_min[kImgsPerPass], row_max[kImgsPerPass], len_max[kImgsPerPass];
    __shared__ uint32_t round_max[3];  // max |G| bits of a round, three slots in rotation (see the reset below)
    __shared__ int nonfinite;
    constexpr int HWS = THREADS / 32;  // half-waves per workgroup

    const int tid = threadIdx.x;
    const int c_base = (NCH == 1) ? blockIdx.y : 0;
    const int v_lo = blockIdx.z * a.band_rows;  // first patch row of this workgroup's band
    const int wg_row = blockIdx.x;
    const int hw = tid >> 5, nhw = HWS, hl = tid & 31;
    const int tile_elems = NCH * a.band_rows * a.pw;
    for (int e = tid; e < tile_elems; e += THREADS) tile[e] = 0ull;
    if (tid < VAA_IMG) bgrid[tid] = base_coord(tid);
    if (tid < 3) round_max[tid] = 0u;
    if (tid == 0) nonfinite = 0;
#ifdef VAA_K2_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long tlast = wall_clock64();
#define K2_STAMP(i) { const long long tn = wall_clock64(); tacc[i] += tn - tlast; tlast = tn; }
#else
#define K2_STAMP(i)
#endif
    int E = kExpUnset;  // exponent of the tile's fixed-point format (workgroup-uniform): quantum = 2^(E + 1 - kCBits)
    int rnd = 0;        // round counter (workgroup-uniform)
    long absorbed = 0;  // footprint pixels added into the tile since the last flush (workgroup-uniform)
    bool flushed = false;

    // tile -> fp32 (this workgroup's partial, or in MULTI mode the image's own gradient), optionally accumulating, then zero the tile
    auto drain = [&](float* dst, int ph, int pw, bool accumulate, bool rezero) {
        // value = tile * quantum, formed as (hi * 2^32 + lo) * 2^q in fp64 (exact), then rounded once to fp32
        const double quantum = (E == kExpUnset) ? 0.0 : __longlong_as_double((long long)(E + 1 - kCBits + 1023) << 52);
        const bool poison = nonfinite != 0;
        const int plane = ph * pw, rows = min(ph, v_lo + a.band_rows) - v_lo;
        const int tp = rows * pw;
        auto value = [&](int t, int cc) {
            const unsigned long long raw = tile[t * NCH + cc];
            const double d = __builtin_fma((double)(int)(raw >> 32), 4294967296.0, (double)(unsigned)raw);
            return (float)(d * quantum);
        };
        const float qnan = __uint_as_float(0x7fc00000u);
        // channel-major walk, four consecutive elements per thread: one 16-byte store per thread where the destination allows it
        const int nq = (tp + 3) >> 2;
        for (int qi = tid; qi < nq * NCH; qi += THREADS) {
            const int cc = (NCH == 1) ? 0 : qi / nq, t0 = (qi - cc * nq) * 4;
            float* o = dst + (size_t)(c_base + cc) * plane + v_lo * pw + t0;
            float v[4];
#pragma unroll
            for (int z = 0; z < 4; ++z) v[z] = (t0 + z < tp) ? value(t0 + z, cc) : 0.0f;
            if (accumulate) {  // only after a flush (workgroup-uniform): the partial already holds earlier images
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    if (t0 + z < tp) v[z] += o[z];
            }
            if (poison) {
#pragma unroll
                for (int z = 0; z < 4; ++z) v[z] = qnan;
            }
            if (t0 + 3 < tp && (reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    if (t0 + z < tp) o[z] = v[z];
            }
        }
        if (rezero) {
            __syncthreads();  // every thread has read `poison`
            if (MULTI && tid == 0) nonfinite = 0;  // per-image output: the next image starts clean (ordered before its phase 1 by the caller's barrier)
            for (int el = tid; el < tp * NCH; el += THREADS) tile[el] = 0ull;
        }
    };

    for (int b0 = wg_row; b0 < a.B; b0 += gx * kImgsPerPass) {
        int nimg = (a.B - b0 + gx - 1) / gx;  // images of this pass: b0, b0+gx, ...
        nimg = nimg < kImgsPerPass ? nimg : kImgsPerPass;
        __syncthreads();  // previous pass done with the tables (also orders the initial zero-fill)
        if (tid < kImgsPerPass) { row_min[tid] = VAA_IMG; row_max[tid] = -1; len_max[tid] = 0; }
        __syncthreads();
        {
        // ---- per-row column bounds of each footprint (conservative) ----
            // threads cover images x 256 row slots (rows 224..255 idle), so a wave never straddles two images and the extent
            // reduction is one shuffle tree + 3 LDS atomics per wave instead of 3 same-address atomics per row
            for (int r0 = 0; r0 < nimg * 256; r0 += THREADS) {
                const int r = r0 + tid;
                const int q = r >> 8, i = r & 255;
                int len = 0;
                if (q < nimg && i < VAA_IMG) {
                    const int b = b0 + q * gx;
                    const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
                    const int ph = MULTI ? a.pdesc[4 * b] : a.ph, pw = MULTI ? a.pdesc[4 * b + 1] : a.pw;
                    int jlo = 0, jhi = -1;
                    if (a.geometry) {
                        float th[6];
    #pragma unroll
                        for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
                        const PixAffine pa = pix_affine(th);
                        // source x must fall in [px-1, px+pw) (corner x0 or x0+1 on the patch); a side that lies on the
                        // frame edge also receives every clamped out-of-frame sample (padding_mode='border').
                        const float xlo = (px == 0) ? -1e30f : (float)(px - 1);
                        const float xhi = (px + pw == VAA_IMG) ? 1e30f : (float)(px + pw);
                        // ... restricted to this workgroup's band of patch rows [v_lo, vb_hi): source y in [py + v_lo - 1, py + vb_hi)
                        const int vb_hi = min(ph, v_lo + a.band_rows);
                        const float ylo = (py == 0 && v_lo == 0) ? -1e30f : (float)(py + v_lo - 1);
                        const float yhi = (py + ph == VAA_IMG && vb_hi == ph) ? 1e30f : (float)(py + vb_hi);
                        float jl = -1e30f, jh = 1e30f;
                        solve_interval(pa.a00, pa.a01 * (float)i + pa.c0, xlo, xhi, jl, jh);
                        solve_interval(pa.a10, pa.a11 * (float)i + pa.c1, ylo, yhi, jl, jh);
                        if (jl <= jh && v_lo < ph) {
                            jlo = (int)fmaxf(0.0f, floorf(jl) - 1.0f);
                            jhi = (int)fminf((float)(VAA_IMG - 1), ceilf(jh) + 1.0f);
                        }
                    } else if (i >= py + v_lo && i < py + min(ph, v_lo + a.band_rows)) {
                        jlo = px;
                        jhi = px + pw - 1;
                    }
                    if (jhi < jlo) { jlo = 0; jhi = -1; }  // an interval wholly beyond the frame (side open to infinity) is an empty row
                    jlo &= ~1;  // a lane's pixel pair starts at an even column: one aligned 4-byte load per plane
                    len = max(0, jhi - jlo + 1);
                    row_word[q][i] = ((uint32_t)jlo << 16) | (uint32_t)len;
                }
                const int wmin = wave_min_i(len > 0 ? i : VAA_IMG), wmax = wave_max_i(len > 0 ? i : -1), wlen = wave_max_i(len);
                if ((tid & 63) == 0 && q < nimg && wlen > 0) {
                    atomicMin(&row_min[q], wmin);
                    atomicMax(&row_max[q], wmax);
                    atomicMax(&len_max[q], wlen);
                }
            }
        }
        __syncthreads();
        K2_STAMP(0)

        for (int q = 0; q < nimg; ++q) {
            const int b = b0 + q * gx;
            const int rmin = row_min[q], nrows = row_max[q] - rmin + 1;
            const int ph = MULTI ? a.pdesc[4 * b] : a.ph, pw = MULTI ? a.pdesc[4 * b + 1] : a.pw;
            const int plane = ph * pw;
            const int v_hi = min(ph, v_lo + a.band_rows);
            if (nrows > 0 && v_lo < ph) {
                const int nseg = (len_max[q] + 63) >> 6;                   // <= 4
                const uint32_t inv_nseg = (65536u + nseg - 1) / nseg;      // exact floor(s/nseg) for s < 9362
                const int nslots = nrows * nseg;
                const int px = a.xy[2 * b], py = a.xy[2 * b + 1];          // workgroup-uniform -> scalar loads
                float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
                if (a.geometry) {
#pragma unroll
                    for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
                }
                const float* pimg = MULTI ? a.patch + a.pdesc[4 * b + 2] : a.patch;
                const uint16_t* gimg = a.g + (size_t)b * 6 * VAA_NPIX;
                const bool ktiled = TILED && HASK && a.keep_t != nullptr;  // workgroup-uniform
                const uint8_t* kimg = (HASK && !ktiled) ? a.keep + (size_t)b * 3 * (VAA_NPIX / 8) : nullptr;  // HASK: K1's keep bits are given
                const uint16_t* ktimg = ktiled ? a.keep_t + (size_t)b * 3 * 256 * kTilePx : nullptr;
                if (!MULTI && absorbed + (long)nslots * 64 > kFlushPixels) {  // int64 headroom (pathological zoom-outs / huge batches only)
                    __syncthreads();
                    drain(a.partial + (size_t)blockIdx.x * 3 * plane, ph, pw, flushed, true);
                    flushed = true;
                    absorbed = 0;
                    __syncthreads();
                }
                absorbed += (long)nslots * 64;

                for (int s0 = 0; s0 < nslots; s0 += nhw * K, ++rnd) {
                    // ======== phase 1: K slots x 2 pixels per thread, every load issued before the first use ========
                    // per pixel a thread keeps {w, n, tile offset, G per channel}; the four corner weights are formed again when it scatters
                    float G[K][2][NCH], fw[K][2], fn[K][2];
                    int t0v[K][2];      // nw corner in tile coordinates, {row - v_lo + 256, col + 256} packed (can be out of range: the validity bits decide)
                    uint32_t flg[K];    // per pixel p (shift 8p): bit 0 inside, 1 uin0, 2 uin1, 3 vin0, 4 vin1, 5..7 kept per channel (no-mask path); bits 16..18 pix0 & 7
                    uint32_t kb[K][NCH], g0[K][NCH], g1[K][NCH];
                    float gt[K][2][NCH];
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int sidx = s0 + hw + k * nhw;
                        flg[k] = 0u;
                        if (s0 + (hw & ~1) + k * nhw >= nslots) continue;  // wave-uniform: neither half-wave has a slot left in this round
                        const int sc = min(sidx, nslots - 1);
                        const int r = (int)(((uint32_t)sc * inv_nseg) >> 16);
                        const int ks = sc - r * nseg;
                        int i, j0, off, len;
                        i = rmin + r;
                        const uint32_t w = row_word[q][i];
                        off = (ks << 6) + 2 * hl; len = (int)(w & 0xffffu);
                        j0 = min((int)(w >> 16) + off, VAA_IMG - 2);  // even
                        const bool lane_in = sidx < nslots && off < len;
                        const int pix0 = i * VAA_IMG + j0;
                        uint32_t f = ktiled ? 0u : (uint32_t)(pix0 & 7) << 16;  // tile-major keep words are shifted to bit 0 when they are loaded
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const int j = j0 + p;
                            int x0 = j, y0 = i;
                            float wf = 0.0f, nf = 0.0f;
                            if (a.geometry) sample_pos_frac(bgrid[j], bgrid[i], th, x0, y0, wf, nf);
                            fw[k][p] = wf; fn[k][p] = nf;
                            const int u0 = x0 - px, v0 = y0 - py;
                            // corners off the patch, off this workgroup's row band or off the frame get weight 0; a pixel without a live corner is skipped
                            const bool uin0 = (unsigned)u0 < (unsigned)pw, uin1 = ((unsigned)(u0 + 1) < (unsigned)pw) && (x0 + 1 < VAA_IMG);
                            const bool vin0 = v0 >= v_lo && v0 < v_hi, vin1 = (v0 + 1 >= v_lo) && (v0 + 1 < v_hi) && (y0 + 1 < VAA_IMG);
                            const bool inside = lane_in && (off + p < len) && (uin0 || uin1) && (vin0 || vin1);
                            uint32_t fp = (inside ? 1u : 0u) | (uin0 ? 2u : 0u) | (uin1 ? 4u : 0u) | (vin0 ? 8u : 0u) | (vin1 ? 16u : 0u);
                            t0v[k][p] = ((min(max(v0 - v_lo, -255), 255) + 256) << 16) | (min(max(u0, -255), 255) + 256);
                            if (!HASK && inside) {  // no stored mask: recompute it from the patch (test / stand-alone use)
                                const Samp s = samp_from_frac(x0, y0, wf, nf);
#pragma unroll
                                for (int cc = 0; cc < NCH; ++cc) {
                                    const int c = c_base + cc;
                                    const float cv = a.geometry ? sample_canvas(pimg + c * plane, ph, pw, px, py, s)
                                                                : canvas_at(pimg + c * plane, ph, pw, px, py, j, i);
                                    if (keep_rule(cv, a.mask_mode)) fp |= 32u << cc;
                                }
                            }
                            f |= fp << (8 * p);
                        }
                        if (TILED && ((f & 0x0101u) != 0u)) {
                            // the lane's two pixels are adjacent elements of one tile row (14 is even): ONE load per channel fetches both
                            // (8 bytes, or 16 when the tile kernel ran one tower per workgroup and left {tower 0, tower 1} pairs).
                            // A pixel without any kept channel may lie in a tile that was not evaluated: its value is read but never used
                            const int ty = i / kTilePx, tx = j0 / kTilePx;
                            const size_t gel = ((size_t)b * (kTilesPerSide * kTilesPerSide) + ty * kTilesPerSide + tx) * kTileElems +
                                               (i - ty * kTilePx) * kTilePx + (j0 - tx * kTilePx);
#pragma unroll
                            for (int cc = 0; cc < NCH; ++cc) {
                                const size_t ge = gel + (c_base + cc) * (kTilePx * kTilePx);  // even
                                if (a.geff2 && a.geff_bf16) {  // one bf16 PLANE per tower (geff, geff2): two adjacent pixels = 4 bytes of each; scaled and added here
                                    const uint32_t w0 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.geff) + ge);
                                    const uint32_t w1 = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(a.geff2) + ge);
                                    const float s0 = a.istd6[c_base + cc], s1 = a.istd6[c_base + cc + 3];
                                    gt[k][0][cc] = bf16_bits_to_f32(w0 & 0xffffu) * s0 + bf16_bits_to_f32(w1 & 0xffffu) * s1;
                                    gt[k][1][cc] = bf16_bits_to_f32(w0 >> 16) * s0 + bf16_bits_to_f32(w1 >> 16) * s1;
                                } else if (a.geff2) {  // {tower 0, tower 1} pairs, added here — the tile kernel's own fp32 sum
                                    const float4 v4 = *reinterpret_cast<const float4*>(a.geff + 2 * ge);
                                    gt[k][0][cc] = v4.x + v4.y;
                                    gt[k][1][cc] = v4.z + v4.w;
                                } else {
                                    const float2 v2 = *reinterpret_cast<const float2*>(a.geff + ge);
                                    gt[k][0][cc] = v2.x;
                                    gt[k][1][cc] = v2.y;
                                }
                            }
                        }
                        flg[k] = f;
                        if (lane_in) {
#pragma unroll
                            for (int cc = 0; cc < NCH; ++cc) {
                                const int c = c_base + cc;
#ifndef VAA_K2_ABLATE_NO_LOADS
                                if (HASK) {
                                    if (ktiled) {  // word (c, tile, y), bit x: a lane's pixel pair never straddles a tile (14 is even)
                                        const int ty = i / kTilePx, tx = j0 / kTilePx;
                                        kb[k][cc] = (uint32_t)ktimg[((size_t)c * 256 + ty * kTilesPerSide + tx) * kTilePx + (i - ty * kTilePx)] >> (j0 - tx * kTilePx);
                                    } else {
                                        kb[k][cc] = kimg[(size_t)c * (VAA_NPIX / 8) + (pix0 >> 3)];
                                    }
                                }
                                if (!TILED) {
                                    g0[k][cc] = *reinterpret_cast<const uint32_t*>(gimg + (size_t)c * VAA_NPIX + pix0);
                                    g1[k][cc] = *reinterpret_cast<const uint32_t*>(gimg + (size_t)(c + 3) * VAA_NPIX + pix0);
                                }
#else
                                kb[k][cc] = 0xffu; g0[k][cc] = 0x3c003c00u + pix0; g1[k][cc] = 0x3c003c00u + c;  // ablation: no global loads
#endif
                            }
                        }
                    }
                    K2_STAMP(1)
                    float lmax = 0.0f;
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const uint32_t fp = flg[k] >> (8 * p);
#pragma unroll
                            for (int cc = 0; cc < NCH; ++cc) G[k][p][cc] = 0.0f;
                            if (s0 + (hw & ~1) + k * nhw >= nslots) continue;  // wave-uniform
#pragma unroll
                            for (int cc = 0; cc < NCH; ++cc) {
                                const int c = c_base + cc;
                                float Gv = 0.0f;
                                if (fp & 1u) {
                                    const bool kept = HASK ? ((kb[k][cc] >> (((flg[k] >> 16) & 7u) + p)) & 1u) : ((fp >> (5 + cc)) & 1u);
                                    // d/d(im) of (im-mean)/std for both normalisations, fp32 exactly like autograd. Reciprocals are
                                    // rounded once on the host (exact for the 0.5 of the second normalisation, <= 1 ulp for the first).
                                    if (kept) {
                                        const float ga = __uint_as_float(p ? (g0[k][cc] & 0xffff0000u) : (g0[k][cc] << 16));
                                        const float gb = __uint_as_float(p ? (g1[k][cc] & 0xffff0000u) : (g1[k][cc] << 16));
                                        Gv = TILED ? gt[k][p][cc] : ga * a.istd6[c] + gb * a.istd6[c + 3];
                                    }
                                }
                                if ((__float_as_uint(Gv) & 0x7fffffffu) >= 0x7f800000u) { bad = true; Gv = 0.0f; }  // inf / nan upstream: poisons the output
                                G[k][p][cc] = Gv;
                                lmax = fmaxf(lmax, fabsf(Gv));
                            }
                        }
                    lmax = wave_max(lmax);
                    if ((tid & 63) == 0 && lmax > 0.0f) atomicMax(&round_max[rnd % 3], __float_as_uint(lmax));
                    if (bad) nonfinite = 1;
                    // slot (rnd+1)%3 was last read after the barrier of round rnd-2, i.e. before every thread arrived at the barrier
                    // of round rnd-1, and is next written after this round's barrier: resetting it here races with neither
                    if (tid == 0) round_max[(rnd + 1) % 3] = 0u;
                    K2_STAMP(2)
                    __syncthreads();
                    K2_STAMP(3)
                    const uint32_t mbits = round_max[rnd % 3];
                    if (mbits == 0u) continue;  // nothing kept in this round (workgroup-uniform)
                    int e = (int)(mbits >> 23) - 127;  // floor(log2(max |G|)) (denormals: -127)
                    e = max(e, -96);                   // keeps 2^(kCBits - 1 - E) a normal float; such gradients are ~1e-29
                    if (E == kExpUnset) {
                        E = e;
                    } else if (e > E) {  // a larger image than any before: re-scale what the tile holds (rare; deterministic)
                        const int d = min(e - E, 62);
                        for (int el = tid; el < tile_elems; el += THREADS) tile[el] = (unsigned long long)shift_round((long long)tile[el], d);
                        E = e;
                        __syncthreads();
                    }
                    // ======== phase 2: integer scatter. contribution = round(fl(G*w) * 2^(29-E)): the reference's fp32 product, to 2^-30 of the
                    //          workgroup's largest |G| (the power-of-two scale commutes with the fp32 rounding of the product) ========
                    const float gscale = __uint_as_float((uint32_t)(kCBits - 1 - E + 127) << 23);  // |G * gscale| < 2^30
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            const uint32_t fp = flg[k] >> (8 * p);
                            if (!(fp & 1u)) continue;  // pixel outside the footprint / the slot: nothing to add
                            const float wf = fw[k][p], nf = fn[k][p];
                            const float ee = 1.0f - wf, so = 1.0f - nf;
                            const float wx0 = (fp & 2u) ? ee : 0.0f, wx1 = (fp & 4u) ? wf : 0.0f;
                            const float wy0 = (fp & 8u) ? so : 0.0f, wy1 = (fp & 16u) ? nf : 0.0f;
                            const float wt[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};  // the fp32 weights grid_sample's backward uses
                            // zero-weight corners (off the patch / band / frame) add 0 to a cell of the pixel's own neighbourhood: the
                            // coordinates are clamped per axis, so lanes keep distinct addresses (a shared dummy cell would serialise)
                            const int vr = (t0v[k][p] >> 16) - 256, uc = (t0v[k][p] & 0xffff) - 256;
                            const int vmaxr = v_hi - v_lo - 1;
                            const int u0c = min(max(uc, 0), pw - 1), u1c = min(max(uc + 1, 0), pw - 1);
                            const int v0c = min(max(vr, 0), vmaxr) * pw, v1c = min(max(vr + 1, 0), vmaxr) * pw;
                            const int offs[4] = {v0c + u0c, v0c + u1c, v1c + u0c, v1c + u1c};
                            unsigned long long* tp[4];
#pragma unroll
                            for (int cn = 0; cn < 4; ++cn) tp[cn] = tile + offs[cn] * NCH;
#pragma unroll
                            for (int cc = 0; cc < NCH; ++cc) {
                                const float Gs = G[k][p][cc] * gscale;
#pragma unroll
                                for (int cn = 0; cn < 4; ++cn) {
                                    const int ci = round_half_up(Gs * wt[cn]);
#ifndef VAA_K2_ABLATE_NO_ATOMICS
                                    atomicAdd(tp[cn] + cc, (unsigned long long)(long long)ci);
#else
                                    if (ci == 0x12345678) tp[cn][cc] = 1ull;  // ablation: keep the arithmetic, drop the LDS atomics
#endif
                                }
                            }
                        }
                    K2_STAMP(4)
                }
            }
            if (MULTI) {  // per-image output: write this image's band of d L / d (its own patch), reset the tile
                __syncthreads();
                if (v_lo < ph) drain(a.partial + a.pdesc[4 * b + 2], ph, pw, false, true);
                __syncthreads();
                E = kExpUnset;
            }
        }
    }
    if (MULTI) return;
    __syncthreads();
    drain(a.partial + (size_t)blockIdx.x * 3 * a.ph * a.pw, a.ph, a.pw, flushed, false);
#ifdef VAA_K2_TIMING
    K2_STAMP(5)
    if ((tid & 63) == 0 && a.geff == nullptr && vaa_k2_dbg)
        for (int z = 0; z < 6; ++z) vaa_k2_dbg[((size_t)blockIdx.x * 16 + (tid >> 6)) * 6 + z] = tacc[z];
#endif
}

// gpatch[e] = sum_p partial[p][e] in a fixed two-level order (16 interleaved slices, then slice 0..15), fp64.
// A thread owns four consecutive elements (16 B loads); workgroup = 16 element-quads x 16 slices (64 elements), so a 3x50x50
// gradient is 118 workgroups and the partial tiles are pulled by that many CUs at once.
__global__ __launch_bounds__(256) void patch_grad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gpatch,
                                                                 int n, int nparts) {
    __shared__ double sl[16][16][4];
    int oe;
    float ov;
    (void)partial_reduce_block(partial, gpatch, n, nparts, blockIdx.x, sl, oe, ov);
}

int launch_partial_reduce(const float* partial, float* gpatch, int n, int nparts, hipStream_t st, const char* who) {
    VAA_LAUNCH(patch_grad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, partial, gpatch, n, nparts);
    return check_launch(who);
}

struct GradSched {
    int gx, bands;
};

// Schedule of the 3-channel path (patches whose int64 tile fits 64 KB). Workgroup x = blockIdx.x owns images x, x + gx, ...; a small batch is
// spread over the chip by cutting the PATCH into row bands (grid.z): a band's workgroup walks only the part of each footprint that maps onto
// its rows, keeps an int64 tile of just those rows and drains just those rows of the partial tile, so the partial buffer stays one tile per
// workgroup-row whatever the band count (slot-interleaved workgroups that each held a full tile were the previous split: 2 x the drain
// and reduce traffic). Measured scatter + reduce, 3x50x50, bands 2 / 4 / 8: B=8 13.9 / 12.1 / 12.0, B=32 16.7 / 14.2 / 13.1,
// B=64 18.3 / 15.1 / 16.9, B=128 23.0 / 21.2 / 29.3 us; from ~130 images on the batch alone fills the chip.
static GradSched grad_sched(int B) {
    GradSched g;
    g.gx = B < 512 ? B : 512;
    g.bands = B <= 32 ? 8 : (B <= 128 ? 4 : 1);
    return g;
}

constexpr size_t kLdsBudget = 144 * 1024;  // 160 KiB per CU minus the static row tables (~9 KB)

#ifndef VAA_K2_TARGET_WGS  // experiment knob
#define VAA_K2_TARGET_WGS 128
#endif

// Patch rows per row band of the one-channel-per-workgroup paths: at most what the LDS holds as an int64 plane, and no more than it
// takes for `wgs_per_band` workgroups x bands to reach ~128 workgroups (a band only walks ITS part of the footprint and drains ITS rows,
// so small batches of large patches are spread over the chip without any extra partial tile). Never below 8 rows.
static int band_rows_for(int ph, int pw, int wgs_per_band) {
    const size_t row_bytes = (size_t)pw * sizeof(long long);
    int rows = (int)(kLdsBudget / row_bytes);
    rows = rows >= ph ? ph : rows;
    const int want = (VAA_K2_TARGET_WGS + wgs_per_band - 1) / wgs_per_band;  // bands wanted
    int r2 = (ph + want - 1) / want;
    r2 = r2 < 8 ? 8 : r2;
    return r2 < rows ? r2 : rows;
}

template <bool TILED>
static int launch_scatter_reduce(const GradArgs& a0, float* gpatch, hipStream_t st, const char* who) {
    GradArgs a = a0;
    const int B = a.B, ph = a.ph, pw = a.pw, n = 3 * ph * pw;
    const GradSched gs = grad_sched(B);
    const int G = gs.gx;  // workgroups (x) == partial tiles
    const size_t plane = (size_t)ph * pw;
    hipError_t e = hipSuccess;
    if (3 * plane * sizeof(long long) <= 64 * 1024) {  // e.g. 50x50: 60,000 B
        // 512 threads: two workgroups per CU (<= 128 VGPRs each, 2 x 68 KB of LDS at one band), so one scatters while the other waits on its
        // loads or its barrier (B = 256 / 1024 / 4096: 41.8 / 76.7 / 183 us with one 1024-thread workgroup per CU -> 33.4 / 64.8 / 173 us)
        a.band_rows = (ph + gs.bands - 1) / gs.bands;
        const int nb = (ph + a.band_rows - 1) / a.band_rows;
        const size_t lds = 3 * (size_t)a.band_rows * pw * sizeof(long long);
        if (a.keep) VAA_LAUNCH((patch_grad_scatter_kernel<3, TILED, false, true, 512, 3>), dim3(G, 1, nb), dim3(512), lds, st, a, gs.gx);
        else VAA_LAUNCH((patch_grad_scatter_kernel<3, TILED, false, false, 512, 3>), dim3(G, 1, nb), dim3(512), lds, st, a, gs.gx);
    } else {  // one channel per workgroup (grid.y), row bands (grid.z) when even one plane exceeds the LDS (> 135x135)
        a.band_rows = band_rows_for(ph, pw, 3 * G);
        const int nbands = (ph + a.band_rows - 1) / a.band_rows;
        const size_t bytes = (size_t)a.band_rows * pw * sizeof(long long);
        const void* fn = a.keep ? (const void*)patch_grad_scatter_kernel<1, TILED, false, true, 1024, 3>
                                : (const void*)patch_grad_scatter_kernel<1, TILED, false, false, 1024, 3>;
        if (bytes > 64 * 1024) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) {
            if (a.keep)
                VAA_LAUNCH((patch_grad_scatter_kernel<1, TILED, false, true, 1024, 3>), dim3(G, 3, nbands), dim3(1024), bytes, st, a, gs.gx);
            else
                VAA_LAUNCH((patch_grad_scatter_kernel<1, TILED, false, false, 1024, 3>), dim3(G, 3, nbands), dim3(1024), bytes, st, a, gs.gx);
        }
    }
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return VAA_E_LAUNCH;
    }
    int rc = check_launch(who);
    if (rc != VAA_OK || !gpatch) return rc;  // gpatch == nullptr: the caller's step epilogue adds the partial tiles
    return launch_partial_reduce((const float*)a.partial, gpatch, n, G, st, who);
}

// one workgroup per (image, channel, row band); every image's own gradient is drained into gpacked
template <bool TILED>
static int launch_scatter_multi(GradArgs a, int max_h, int max_w, hipStream_t st, const char* who) {
    const int B = a.B;
    a.band_rows = band_rows_for(max_h, max_w, 3 * B);
    const int nbands = (max_h + a.band_rows - 1) / a.band_rows;
    const size_t bytes = (size_t)a.band_rows * max_w * sizeof(long long);
    const void* fn = a.keep ? (const void*)patch_grad_scatter_kernel<1, TILED, true, true, 1024, 3>
                            : (const void*)patch_grad_scatter_kernel<1, TILED, true, false, 1024, 3>;
    if (bytes > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        set_error("%s: hipFuncSetAttribute failed", who);
        return VAA_E_LAUNCH;
    }
    if (a.keep)
        VAA_LAUNCH((patch_grad_scatter_kernel<1, TILED, true, true, 1024, 3>), dim3(B, 3, nbands), dim3(1024), bytes, st, a, B);
    else
        VAA_LAUNCH((patch_grad_scatter_kernel<1, TILED, true, false, 1024, 3>), dim3(B, 3, nbands), dim3(1024), bytes, st, a, B);
    return check_launch(who);
}

}  // namespace vaa

#ifdef VAA_K2_TIMING
extern "C" int vaa_k2_set_debug(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(vaa::vaa_k2_dbg), &p, sizeof(p)); }
#endif

extern "C" size_t vaa_patch_grad_ws_bytes(int B, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0) return 0;
    const vaa::GradSched g = vaa::grad_sched(B);
    return (size_t)g.gx * 3 * (size_t)ph * pw * sizeof(float);
}

extern "C" int vaa_patch_grad_gather(const uint16_t* gout_bf16, const float* patch, const int32_t* xy, const float* theta,
                                     const uint8_t* keep_bits, int B, int ph, int pw, int geometry, int mask_mode,
                                     const float* std6, float* gpatch, void* ws, size_t ws_bytes, void* stream) {
    using namespace vaa;
    if (B == 0 && gpatch && ph > 0 && pw > 0) {  // empty batch: the sum over no images is zero
        if (hipMemsetAsync(gpatch, 0, (size_t)3 * ph * pw * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return check_launch("vaa_patch_grad_gather(memset)");
        return VAA_OK;
    }
    if (!gout_bf16 || !xy || !std6 || (geometry && !theta) || (!keep_bits && !patch)) {  // gpatch == NULL: the final sum is left to vaa_step_epilogue
        set_error("vaa_patch_grad_gather: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0 || (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("vaa_patch_grad_gather: bad sizes/mode (B=%d ph=%d pw=%d mask_mode=%d)", B, ph, pw, mask_mode);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("vaa_patch_grad_gather: patch %dx%d larger than the frame", ph, pw);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        set_error("vaa_patch_grad_gather: VAA_MASK_NE_M100 is defined for geometry=0 only");
        return VAA_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (!ws || ws_bytes < vaa_patch_grad_ws_bytes(B, ph, pw)) {
        set_error("vaa_patch_grad_gather: workspace %zu B < required %zu B", ws_bytes, vaa_patch_grad_ws_bytes(B, ph, pw));
        return VAA_E_WORKSPACE;
    }
    GradArgs a;
    a.g = gout_bf16; a.patch = patch; a.xy = xy; a.theta = theta; a.keep = keep_bits; a.partial = (float*)ws; a.pdesc = nullptr;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode; a.band_rows = ph;
    for (int q = 0; q < 6; ++q) a.istd6[q] = (float)(1.0 / (double)std6[q]);
    a.geff = nullptr; a.geff2 = nullptr; a.geff_bf16 = 0; a.keep_t = nullptr;
    return launch_scatter_reduce<false>(a, gpatch, st, "vaa_patch_grad_gather");
}

// K2 with one patch per image (resize_patch=True, appply_random_transform.py:113-118): image b's patch is packed + offset_b,
// [3,h_b,w_b]; gpacked receives d L / d (that patch) in the same layout. One workgroup per (image, channel, row band).
extern "C" int vaa_patch_grad_gather_multi(const uint16_t* gout_bf16, const float* packed, const int32_t* pdesc, const int32_t* xy,
                                           const float* theta, const uint8_t* keep_bits, int B, int max_h, int max_w, int geometry,
                                           int mask_mode, const float* std6, float* gpacked, void* stream) {
    using namespace vaa;
    if (B == 0) return VAA_OK;
    if (!gout_bf16 || !pdesc || !xy || !std6 || !gpacked || (geometry && !theta) || (!keep_bits && !packed)) {
        set_error("vaa_patch_grad_gather_multi: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B < 0 || max_h <= 0 || max_w <= 0 || (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("vaa_patch_grad_gather_multi: bad sizes/mode (B=%d max_h=%d max_w=%d mask_mode=%d)", B, max_h, max_w, mask_mode);
        return VAA_E_INVALID;
    }
    if (max_h > VAA_IMG || max_w > VAA_IMG) {
        set_error("vaa_patch_grad_gather_multi: patch bound %dx%d larger than the frame", max_h, max_w);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        set_error("vaa_patch_grad_gather_multi: VAA_MASK_NE_M100 is defined for geometry=0 only");
        return VAA_E_UNSUPPORTED;
    }
    GradArgs a;
    a.g = gout_bf16; a.patch = packed; a.xy = xy; a.theta = theta; a.keep = keep_bits; a.partial = gpacked; a.pdesc = pdesc;
    a.B = B; a.ph = max_h; a.pw = max_w; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) a.istd6[q] = (float)(1.0 / (double)std6[q]);
    a.geff = nullptr; a.geff2 = nullptr; a.geff_bf16 = 0; a.keep_t = nullptr;
    return launch_scatter_multi<false>(a, max_h, max_w, (hipStream_t)stream, "vaa_patch_grad_gather_multi");
}

// ------------------------------------------------------------------------------------------------------------------------------
// the ViT patch-embed backward restricted to the 14x14 tiles that carry kept patch pixels, fused with the gather.
// The patch-embed conv (kernel == stride == 14) is a GEMM over 588-pixel tiles (modeling_prismatic.py:120-123 evaluates timm's
// PatchEmbed of both towers), so dL/dpixel of one tile is dY[tile, :] @ W[:, 588]. Only ~36 of the 256 tiles of an image are under
// the warped patch; their gradients are produced by MFMA straight into a compact fp32 buffer (both towers combined and divided by
// the two normalisation stds) that the scatter kernel above reads in place of the 6-plane bf16 pixel gradient.
namespace vaa {

typedef short v8s_e __attribute__((ext_vector_type(8)));
typedef float v4f_e __attribute__((ext_vector_type(4)));

struct EmbedArgs {
    const uint16_t *dy0, *dy1;  // [B,256,D0], [B,256,D1] bf16: dL/d(patch-embed output) of the two towers, tokens in tile order
    const uint16_t *wt0, *wt1;  // conv weights of the two towers in the PACKED fragment order of embed_pack_weights_kernel
    const uint8_t* keep;        // [B,3,224*224/8] keep bits from K1
    const uint32_t* flags;      // [B,256] tile flag words from K1's tile-major form (else nullptr: derived from `keep`)
    float* geff;                // [B,256,588], indexed by tile (ty*16 + tx); only the flagged tiles are written
    float* geff2;               // non-null: room for the pair layout of tower_split ({tower 0, tower 1} per element, in geff)
    int B, D0, D1, round_bf16;
    int tower_split;            // one tower per workgroup: halves the per-workgroup chain while the launch is far from filling the chip
    int pair_bf16;              // tower_split with round_bf16: geff / geff2 are bf16 planes of the towers' rounded values, unscaled (the gather scales and adds)
    int ny;                     // 64-tile row groups of an image that go to separate workgroups (1: a workgroup walks them in sequence)
    float istd6[6];
};

constexpr int kNBlocks = (kTileElems + 15) / 16;  // 37 column blocks of 16 pixels-of-a-tile

__device__ __forceinline__ v8s_e frag_or_zero(const uint16_t* p, bool valid) {
    uint4 r = make_uint4(0, 0, 0, 0);
    if (valid) r = *reinterpret_cast<const uint4*>(p);
    return *reinterpret_cast<v8s_e*>(&r);
}

__device__ __forceinline__ float maybe_bf16(float v, int on) { return on ? bf16_bits_to_f32(f32_to_bf16_bits(v)) : v; }

// Packed weight layout (built once per model by vaa_patch_embed_pack_weights; the weights are frozen during an attack):
//   packed[(((nb * nchunk + kc) * 2 + h) * 64 + lane) * 8 + e] = W^T[n = nb*16 + (lane & 15)][k = kc*64 + (lane >> 4)*16 + h*8 + e]   (0 for n >= 588)
// nb = 16-column block (37 of them), kc = 64-wide k-chunk, h = k-half: exactly the B operand of one mfma_f32_16x16x32_bf16, so a wave's
// fragment load is ONE contiguous KB (8 full lines). Reading the fragments straight from the [588,D] matrix made every load instruction
// touch 16 lines at a 2 KB stride: the k-loops ran at ~20 B/clk per CU (12 + 14 us for the two towers against 4 + 4 us of MFMA/LDS time).
__device__ __forceinline__ size_t packed_frag_offset(int nb, int nchunk, int kc, int h, int lane) {
    return ((((size_t)nb * nchunk + kc) * 2 + h) * 64 + lane) * 8;
}

__global__ __launch_bounds__(256) void embed_pack_weights_kernel(const uint16_t* __restrict__ wt, int D, uint16_t* __restrict__ packed) {
    const int nchunk = D >> 6;
    const long nfrag = (long)kNBlocks * nchunk * 2 * 64;
    for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < nfrag; f += (long)gridDim.x * 256) {
        const int lane = (int)(f & 63), h = (int)((f >> 6) & 1);
        const long t = f >> 7;
        const int kc = (int)(t % nchunk), nb = (int)(t / nchunk);
        const int n = nb * 16 + (lane & 15), k = kc * 64 + (lane >> 4) * 16 + h * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < kTileElems) v = *reinterpret_cast<const uint4*>(wt + (size_t)n * D + k);
        *reinterpret_cast<uint4*>(packed + f * 8) = v;
    }
}

// Does tile (ty, tx) hold a kept pixel in any channel? 3 planes x 14 rows x 14 bits, read as 126 INDEPENDENT byte loads that are all in
// flight together (an early-exit loop makes them 42 dependent round trips: it was 35 of the kernel's 57 us).
__device__ __forceinline__ bool tile_has_kept_pixel(const uint8_t* kb, int ty, int tx) {
    uint32_t any = 0u;
#pragma unroll
    for (int ch3 = 0; ch3 < 3; ++ch3) {
        const uint8_t* p = kb + (size_t)ch3 * (VAA_NPIX / 8);
#pragma unroll
        for (int y = 0; y < kTilePx; ++y) {
            const int bit0 = (ty * kTilePx + y) * VAA_IMG + tx * kTilePx;  // 14 consecutive bits: within one (unaligned) 32-bit word
            const int by = min(bit0 >> 3, VAA_NPIX / 8 - 4);              // the last word of a plane is read 1..3 bytes early
            uint32_t w;
            __builtin_memcpy(&w, p + by, 4);
            any |= (w >> (bit0 - 8 * by)) & 0x3fffu;
        }
    }
    return any != 0u;
}

constexpr int kEmbedThreads = 256;

// grid = B * nch workgroups of 4 waves; workgroup (b, ch) evaluates column blocks ch*4 .. ch*4+3 (one per wave) for every flagged tile
// of image b. (Splitting the contraction over 16 waves with an LDS reduction was measured slower: 121 vs 90 us at bs=64.)
__global__ __launch_bounds__(kEmbedThreads) void embed_dgrad_tiles_kernel(EmbedArgs a, int nch) {
    __shared__ int16_t tiles[256];
    __shared__ int wave_cnt[4];
    // all workgroups of an image share blockIdx % 8, i.e. one XCD and one L2: its dY rows are then fetched from HBM once instead of
    // once per XCD (PMC: 82 MB -> see profiles/traffic_r01.json)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int b = (slot_id / nch) * 8 + xcd, ch = slot_id % nch;
    if (b >= a.B) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;

    // ---- which tiles carry a kept pixel (any channel): thread t = tile t ----
    const bool flag = a.flags ? a.flags[(size_t)b * 256 + tid] != 0 : tile_has_kept_pixel(a.keep + (size_t)b * 3 * (VAA_NPIX / 8), tid >> 4, tid & 15);
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int base = 0, M = 0;
    for (int q = 0; q < 4; ++q) { if (q < wv) base += wave_cnt[q]; M += wave_cnt[q]; }
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (flag) tiles[slot] = (int16_t)tid;
    __syncthreads();

    const int nb = ch * 4 + wv;
    if (nb >= kNBlocks) return;
    const int n = nb * 16 + c;                  // this lane's column: element n of a tile (c3*196 + y*14 + x)
    const bool nv = n < kTileElems;
    const int c3 = nv ? n / (kTilePx * kTilePx) : 0;
    const float s0 = a.istd6[c3], s1 = a.istd6[c3 + 3];
    // Up to four 16-tile row blocks share every weight fragment: per 64-wide k-chunk 2 B-fragments + 8 A-fragments are in flight, 8 MFMAs follow.
    for (int mg = blockIdx.y; mg * 64 < M; mg += gridDim.y) {  // 64-tile row groups: spread over grid.y when the patch can cover more
        v4f_e acc0[4], acc1[4];
        const uint16_t *y0[4], *y1[4];
        bool mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0[q] = (v4f_e){0.f, 0.f, 0.f, 0.f};
            acc1[q] = (v4f_e){0.f, 0.f, 0.f, 0.f};
            const int mr = mg * 64 + q * 16 + c;  // A-operand row of this lane in row block q
            mv[q] = mr < M;
            const int tile = mv[q] ? tiles[mr] : 0;
            y0[q] = a.dy0 + ((size_t)b * 256 + tile) * a.D0 + g * 16;  // lane group g owns k = 16g .. 16g+15 of a chunk (see the packed layout)
            y1[q] = a.dy1 + ((size_t)b * 256 + tile) * a.D1 + g * 16;
        }
        const int nq = min(4, (M - mg * 64 + 15) / 16);  // live row blocks (wave-uniform)
#pragma unroll
        for (int tower = 0; tower < 2; ++tower) {
            const int D = tower ? a.D1 : a.D0, nchunk = D >> 6;
            const uint16_t* wp = (tower ? a.wt1 : a.wt0) + packed_frag_offset(nb, nchunk, 0, 0, lane);
            for (int kc = 0; kc < nchunk; ++kc) {
                const v8s_e b0 = *reinterpret_cast<const v8s_e*>(wp + (size_t)kc * 1024), b1 = *reinterpret_cast<const v8s_e*>(wp + (size_t)kc * 1024 + 512);
                v8s_e a0[4], a1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint16_t* yp = tower ? y1[q] : y0[q];
                    a0[q] = frag_or_zero(yp + kc * 64, mv[q]);
                    a1[q] = frag_or_zero(yp + kc * 64 + 8, mv[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nq) continue;
                    if (tower) {
                        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[q], b0, acc1[q], 0, 0, 0);
                        acc1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[q], b1, acc1[q], 0, 0, 0);
                    } else {
                        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[q], b0, acc0[q], 0, 0, 0);
                        acc0[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[q], b1, acc0[q], 0, 0, 0);
                    }
                }
            }
        }
        if (nv) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // C/D layout: column = lane&15 (pixel n), row = 4*(lane>>4)+r (tile slot)
                    const int sl = mg * 64 + q * 16 + g * 4 + r;
                    if (sl < M)
                        a.geff[((size_t)b * 256 + tiles[sl]) * kTileElems + n] =
                            maybe_bf16(acc0[q][r], a.round_bf16) * s0 + maybe_bf16(acc1[q][r], a.round_bf16) * s1;
                }
        }
    }
}


#ifndef VAA_EMBED_GROUP
#define VAA_EMBED_GROUP 2
#endif
constexpr int kEmbedGroup = VAA_EMBED_GROUP;  // 64-wide k-chunks whose weight fragments a wave requests together (16 VGPRs each)

// The barrier-free k-loop of the fast variant for NQ live 16-row blocks. K is walked in chunks of 64: lane group g owns k = 16g .. 16g+15
// of a chunk, the first eight for one MFMA, the next eight for a second (any assignment of k to lanes is valid as long as A and B agree);
// wp0 / wp1 point at this lane's slot of chunk 0 in the packed weights of the wave's two column blocks.
//   * the weight fragments of kEmbedGroup chunks are requested together, UNCONDITIONALLY (the last groups re-request the final chunk), into two register sets in ping-pong, and consumed in order under counting waits. With
//     predicated loads, or a ring refilled chunk by chunk, the compiler falls back to s_waitcnt vmcnt(0) in front of every chunk: one
//     full L2 round trip per chunk, 13 us per tower.
//   * the A fragments of chunk u+1 are read from LDS before the MFMAs of chunk u (one ds_read per row block and k-half, all issued
//     together), so the LDS latency is covered by twelve MFMAs instead of being paid in front of every pair.
//   * MFMA order: all row blocks against the first k-half, then the second: consecutive MFMAs never share an accumulator.
template <int NQ, int NB, int G = kEmbedGroup>
__device__ __forceinline__ void embed_kloop(const uint16_t* ap, int SA, const uint16_t* const (&wp)[NB], int nchunk, v4f_e (&acc)[NB][4],
                                            const v8s_e (*pre)[NB][2] = nullptr) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = (v4f_e){0.f, 0.f, 0.f, 0.f};
    auto load_group = [&](v8s_e (&bf)[G][NB][2], int k0) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int kc = min(k0 + u, nchunk - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NB; ++j) bf[u][j][h] = *reinterpret_cast<const v8s_e*>(wp[j] + (size_t)kc * 1024 + h * 512);
        }
    };
    auto compute_group = [&](const v8s_e (&bf)[G][NB][2], int k0) {
        if (k0 >= nchunk) return;  // wave-uniform
        v8s_e af[2][NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) af[0][q][h] = *reinterpret_cast<const v8s_e*>(ap + q * 16 * SA + k0 * 64 + h * 8);
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int kc = k0 + u;
            if (kc < nchunk) {  // wave-uniform
                if (u + 1 < G) {
                    const int kn = min(kc + 1, nchunk - 1);
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int h = 0; h < 2; ++h) af[(u + 1) & 1][q][h] = *reinterpret_cast<const v8s_e*>(ap + q * 16 * SA + kn * 64 + h * 8);
                }
                __builtin_amdgcn_sched_barrier(0);  // the next chunk's LDS reads are issued before this chunk's MFMAs
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u & 1][q][h], bf[u][j][h], acc[j][q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // two register sets in ping-pong: the requests of the next group are in flight while this group's MFMAs run
    v8s_e bfa[G][NB][2], bfb[G][NB][2];
    if (pre) {  // the first group was requested by the caller, under its staging round trip
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) bfa[u][j][h] = pre[u][j][h];
    } else {
        load_group(bfa, 0);
    }
    for (int k0 = 0; k0 < nchunk; k0 += 2 * G) {
        load_group(bfb, k0 + G);
        __builtin_amdgcn_sched_barrier(0);
        compute_group(bfa, k0);
        load_group(bfa, k0 + 2 * G);
        __builtin_amdgcn_sched_barrier(0);
        compute_group(bfb, k0 + G);
    }
}

// Fast variant for tower widths that fit the LDS (64 x (D+8) bf16 <= 150 KB, i.e. D <= 1160): the gathered dY rows of one tower are
// staged ONCE per workgroup (all loads in flight together, one barrier), then every wave runs a barrier-free k-loop — A fragments from
// LDS (16 B reads, each feeding NB MFMAs: a wave owns NB column blocks), B fragments (weights) from global, four k-steps ahead.
// The 37 column blocks of a tile are cut into `nch` contiguous ranges (one workgroup each, all on the image's XCD) and a range is dealt
// to the 8 waves as evenly as possible (1..3 blocks per wave, the waves with one more block on different SIMDs).
#ifndef VAA_EMBED_WAVES
#define VAA_EMBED_WAVES 8
#endif
constexpr int kEmbedFastThreads = VAA_EMBED_WAVES * 64;
constexpr int kEmbedStageMax = ((64 * 1160 / 8 + kEmbedFastThreads - 1) / kEmbedFastThreads + 1) / 2 * 2;  // 16-byte chunks a thread stages per tower (even)

// SPLIT: one tower per workgroup (blockIdx.z), its tile gradients into geff / geff2 (the gather adds the two); else both towers, summed.
// NB: most column blocks a wave owns (ceil(ceil(37 / nch) / 8)).
template <bool SPLIT, int NB>
__global__ __launch_bounds__(kEmbedFastThreads) void embed_dgrad_tiles_lds_kernel(EmbedArgs a, int nch) {
    extern __shared__ __align__(16) unsigned char embed_smem[];
    uint16_t* sA = reinterpret_cast<uint16_t*>(embed_smem);
    __shared__ int16_t tiles[256];
    __shared__ int wave_cnt[4];
    // A UNIT of work = (image b, 64-tile row group mg0 of ny, tower): its nch column-range workgroups read the same dY rows, so they share
    // blockIdx.x % 8 = one XCD and one L2 (the rows are fetched from HBM once); different units touch disjoint rows and are dealt round-robin
    // over the 8 XCDs — unit = (tower * ny + mg0) * B + b, so that a full batch keeps image b on XCD b % 8 for both towers, while a small batch
    // with large per-image patches (resize_patch: 4 images x 4 row groups x 2 towers) still covers all 8 XCDs instead of B of them.
    // SPLIT: a tower's units go to ONE half of the XCDs (tower 0: XCDs 0-3, tower 1: 4-7), so an L2 is filled with one tower's packed weights,
    // not both (8 x 2.56 MB -> 4 x 1.21 + 4 x 1.36 MB of fills at the OpenVLA widths: profiles/traffic_r05.json had the kernel fetch 27.2 MB
    // for 5.8 MB of dY rows)
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int ch = slot_id % nch, ny = a.ny;
    int unit;
    if (SPLIT) {
        const int per_tower = a.B * ny, v = (slot_id / nch) * 4 + (xcd & 3);
        if (v >= per_tower) return;
        unit = (xcd >> 2) * per_tower + v;
    } else {
        unit = (slot_id / nch) * 8 + xcd;
        if (unit >= a.B * ny) return;
    }
    const int b = unit % a.B, mg0 = (unit / a.B) % ny;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, c = lane & 15, g = lane >> 4;
#ifdef VAA_K2_TIMING
    long long tacc[6] = {0, 0, 0, 0, 0, 0};
    long long tlast = wall_clock64();
#endif

    bool flag = false;
    if (tid < 256) flag = a.flags ? a.flags[(size_t)b * 256 + tid] != 0 : tile_has_kept_pixel(a.keep + (size_t)b * 3 * (VAA_NPIX / 8), tid >> 4, tid & 15);
    const unsigned long long m = __ballot(flag);
    if (lane == 0 && wv < 4) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int base = 0, M = 0;
    for (int q = 0; q < 4; ++q) { if (q < wv) base += wave_cnt[q]; M += wave_cnt[q]; }
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (flag) tiles[slot] = (int16_t)tid;
    __syncthreads();

    K2_STAMP(0)
    // this wave's column blocks: [nb0, nb0 + nbw) of the workgroup's range [ch * per, min(37, (ch + 1) * per))
    const int per = (kNBlocks + nch - 1) / nch;
    const int wg_lo = ch * per, wg_n = max(0, min(kNBlocks, wg_lo + per) - wg_lo);
    const int wbase = wg_n / VAA_EMBED_WAVES, wrem = wg_n - wbase * VAA_EMBED_WAVES;
    const int nbw = wbase + (wv < wrem ? 1 : 0), nb0 = wg_lo + wv * wbase + min(wv, wrem);  // nbw <= NB by the host's choice of NB
    int n[NB];
    bool nv[NB];
    float s0[NB], s1[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        n[j] = (nb0 + j) * 16 + c;
        nv[j] = j < nbw && n[j] < kTileElems;
        const int c3 = nv[j] ? n[j] / (kTilePx * kTilePx) : 0;
        s0[j] = a.istd6[c3];
        s1[j] = a.istd6[c3 + 3];
    }
    for (int mg = mg0; mg * 64 < M; mg += ny) {  // 64-tile row groups: separate units when the patch can cover more than 64 tiles
        const int rows = min(64, M - mg * 64), nq = (rows + 15) >> 4;  // workgroup-uniform
        float res[NB][4][4];  // [column block][row block][r]: tower 0's scaled contribution, then + tower 1's
        const int t_lo = SPLIT ? unit / (a.B * ny) : 0;
#pragma unroll
        for (int tt = 0; tt < (SPLIT ? 1 : 2); ++tt) {
            const int tower = SPLIT ? t_lo : tt;
            const int D = tower ? a.D1 : a.D0, SA = D + 8, cpr = D >> 3;  // 16-byte chunks per row
            const uint16_t* dy = tower ? a.dy1 : a.dy0;
            const int nchunks = nq * 16 * cpr;
            // ---- stage: the loads of a phase are all in flight before its LDS stores. One tower per workgroup (SPLIT) has the registers for
            //      ONE phase (20 x 16 B per thread: one memory round trip instead of two); both towers in sequence keep two halves ----
            __syncthreads();  // the previous tower's k-loops are over
            constexpr int kPhases = SPLIT ? 1 : 2, kPer = kEmbedStageMax / kPhases;
            const uint16_t* wt = tower ? a.wt1 : a.wt0;
            // blocks beyond this wave's count (or beyond the 37th) re-read its last block: loads stay unconditional, results are never stored
            const uint16_t* wp[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                wp[j] = wt + packed_frag_offset(min(nb0 + min(j, max(nbw - 1, 0)), kNBlocks - 1), D >> 6, 0, 0, lane);
            // the k-loop's first weight group is requested right behind the staging loads — one memory round trip for both instead of two
            // (B=24: 11.6 -> 10.6 us warm, 14.7 -> 12.7 us with cold caches, profiles/r03_cold_probe.txt)
            constexpr bool kEarly = SPLIT;
            // weight fragments requested together per k-loop group: two 64-wide chunks where a wave owns <= 2 column blocks; ONE with three
            // blocks (the second chunk's 24 VGPRs per register set are what spilled: B=64 18.1 -> 16.9 us warm, 21.7 -> 19.7 us cold)
            constexpr int GRP = (SPLIT && NB >= 3) ? 1 : kEmbedGroup;
            v8s_e pre[GRP][NB][2];
#pragma unroll
            for (int hf = 0; hf < kPhases; ++hf) {
                uint4 st[kPer];
#pragma unroll
                for (int it = 0; it < kPer; ++it) {
                    int idx = tid + (hf * kPer + it) * kEmbedFastThreads;
                    // opaque to the optimiser: otherwise the row / column split of all staging slots is hoisted out of the row-group
                    // loop, stays live across both k-loops and spills (every staging load then sat between two scratch accesses)
                    asm volatile("" : "+v"(idx));
                    st[it] = make_uint4(0, 0, 0, 0);
                    if (idx < nchunks) {
                        const int row = idx / cpr, cc = idx - row * cpr, r = mg * 64 + row;
                        if (r < M) st[it] = *reinterpret_cast<const uint4*>(dy + ((size_t)b * 256 + tiles[r]) * D + cc * 8);
                    }
                }
                if constexpr (kEarly) {
#pragma unroll
                    for (int u = 0; u < GRP; ++u) {
                        const int kc = min(u, (D >> 6) - 1);
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int j = 0; j < NB; ++j) pre[u][j][h] = *reinterpret_cast<const v8s_e*>(wp[j] + (size_t)kc * 1024 + h * 512);
                    }
                }
#pragma unroll
                for (int it = 0; it < kPer; ++it) {
                    int idx = tid + (hf * kPer + it) * kEmbedFastThreads;
                    asm volatile("" : "+v"(idx));
                    if (idx < nchunks) {
                        const int row = idx / cpr, cc = idx - row * cpr;
                        *reinterpret_cast<uint4*>(&sA[row * SA + cc * 8]) = st[it];
                    }
                }
            }
            __syncthreads();
            K2_STAMP(1 + 2 * tower)
            // ---- barrier-free k-loop ----
            v4f_e acc[NB][4];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] = (v4f_e){0.f, 0.f, 0.f, 0.f};
            const uint16_t* ap = &sA[c * SA + g * 16];
            // a wave with fewer blocks than NB runs the narrower loop (wave-uniform): no dummy MFMAs on the SIMD it shares with a fuller wave
            auto run = [&](auto nbtag) {
                constexpr int NBW = decltype(nbtag)::value;
                const uint16_t* wq[NBW];
                v4f_e ac[NBW][4];
                v8s_e pq[GRP][NBW][2];
#pragma unroll
                for (int j = 0; j < NBW; ++j) {
                    wq[j] = wp[j];
                    if constexpr (kEarly) {
#pragma unroll
                        for (int u = 0; u < GRP; ++u)
#pragma unroll
                            for (int h = 0; h < 2; ++h) pq[u][j][h] = pre[u][j][h];
                    }
                }
                const v8s_e (*pp)[NBW][2] = kEarly ? pq : nullptr;
                switch (nq) {
                    case 1: embed_kloop<1, NBW, GRP>(ap, SA, wq, D >> 6, ac, pp); break;
                    case 2: embed_kloop<2, NBW, GRP>(ap, SA, wq, D >> 6, ac, pp); break;
                    case 3: embed_kloop<3, NBW, GRP>(ap, SA, wq, D >> 6, ac, pp); break;
                    default: embed_kloop<4, NBW, GRP>(ap, SA, wq, D >> 6, ac, pp); break;
                }
#pragma unroll
                for (int j = 0; j < NBW; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j][q] = ac[j][q];
            };
            if (nbw >= NB) run(std::integral_constant<int, NB>{});
            else if (NB >= 3 && nbw == 2) run(std::integral_constant<int, (NB >= 3 ? 2 : 1)>{});
            else if (NB >= 2 && nbw == 1) run(std::integral_constant<int, 1>{});
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (SPLIT && a.pair_bf16) {  // the rounded value itself (as a float): the gather scales and adds
                            res[j][q][r] = bf16_bits_to_f32(f32_to_bf16_bits(acc[j][q][r]));
                        } else {
                            const float v = maybe_bf16(acc[j][q][r], a.round_bf16) * (tower ? s1[j] : s0[j]);
                            res[j][q][r] = (tower != t_lo) ? res[j][q][r] + v : v;
                        }
                    }
            K2_STAMP(2 + 2 * tower)
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (!nv[j]) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int sl = mg * 64 + q * 16 + g * 4 + r;
                    if (sl < M) {
                        const size_t el = ((size_t)b * 256 + tiles[sl]) * kTileElems + n[j];
                        if (SPLIT && a.pair_bf16)  // a bf16 plane per tower (the towers run on different XCD halves: interleaved halves of one word were written back by both L2s)
                            reinterpret_cast<uint16_t*>(t_lo ? a.geff2 : a.geff)[el] = (uint16_t)(__float_as_uint(res[j][q][r]) >> 16);  // (exact: rounded above)
                        else if (SPLIT) a.geff[el * 2 + t_lo] = res[j][q][r];  // {tower 0, tower 1} interleaved: ONE 8-byte load per element in the gather
                        else a.geff[el] = res[j][q][r];
                    }
                }
        }
        K2_STAMP(5)
    }
#ifdef VAA_K2_TIMING  // tools/scratch/k2etiming.py: flags / stage 0 / k-loop 0 / stage 1 / k-loop 1 / store, per wave
    if (lane == 0 && vaa_k2_dbg)
        for (int z = 0; z < 6; ++z) vaa_k2_dbg[((size_t)blockIdx.x * 16 + wv) * 6 + z] = tacc[z];
#endif
}

}  // namespace vaa

namespace vaa {

// tile gradients of every image (both towers) into e.geff; the LDS-resident variant while a tower's 64 gathered rows fit
static int launch_embed_tiles(EmbedArgs& e, int ph, int pw, hipStream_t st, const char* who) {
    const int B = e.B, Dmax = e.D0 > e.D1 ? e.D0 : e.D1;
    // a warped ph x pw patch touches at most ~(1.5 ph / 14 + 2) x (1.5 pw / 14 + 2) of the 256 tiles of a frame; beyond 64 tiles the row
    // groups of an image go to separate workgroups (grid.y) instead of being walked in sequence (B=4, 100..139 px: 40 -> 21 us)
    const int tiles_bound = ((3 * ph / 2 + 13) / 14 + 2) * ((3 * pw / 2 + 13) / 14 + 2);
    const unsigned ny = (unsigned)(tiles_bound <= 64 ? 1 : (tiles_bound <= 128 ? 2 : 4));
    const size_t lds_fast = (size_t)64 * (Dmax + 8) * sizeof(uint16_t);
    e.tower_split = 0;
    e.pair_bf16 = 0;
    e.ny = (int)ny;
    if (lds_fast <= 150 * 1024 && (size_t)64 * (Dmax / 8) <= (size_t)kEmbedStageMax * kEmbedFastThreads) {
        // Workgroups per unit (image, row group). A workgroup's time is a chain — tile list, staging of a tower's rows, k-loop, (second tower), store —,
        // so while the launch fits ONE residency wave of the 256 CUs a workgroup takes ONE tower (its chain halves; the gather adds the two towers'
        // tile gradients — the very fp32 add the unsplit kernel does) and as few column blocks as that leaves room for:
        //   <= 24 (image, row group) units: 5 x 2 workgroups each, 1 block per wave;  <= 40: 3 x 2, <= 2 blocks;  <= 64: 2 x 2, <= 3 blocks (256 workgroups at 64)
        // beyond that the batch fills the chip several times over and the unsplit form (3 per unit, both towers, no second buffer) is kept.
        // Per-image patches (resize_patch, ny > 1) count B * ny units: round 4 ran them unsplit on B of the 8 XCDs (B=4, 61..139 px: 31-40 us).
        const long Bpad = (B + 7) / 8 * 8;
        const long eff = ny == 1 ? Bpad : (long)B * ny;
        int nch = 3;
        if (e.geff2 && eff <= 64) {
            e.tower_split = 1;
            nch = eff <= 24 ? 5 : (eff <= 40 ? 3 : 2);
        }
        const int nbmax = ((kNBlocks + nch - 1) / nch + VAA_EMBED_WAVES - 1) / VAA_EMBED_WAVES;  // 1, 2 or 3
        const void* fn = nullptr;
        if (e.tower_split) fn = nbmax == 1 ? (const void*)embed_dgrad_tiles_lds_kernel<true, 1> : (nbmax == 2 ? (const void*)embed_dgrad_tiles_lds_kernel<true, 2> : (const void*)embed_dgrad_tiles_lds_kernel<true, 3>);
        else fn = (const void*)embed_dgrad_tiles_lds_kernel<false, 2>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast) != hipSuccess) {
            set_error("%s: hipFuncSetAttribute failed", who);
            return VAA_E_LAUNCH;
        }
        const long units = (long)B * ny;
        // SPLIT: every XCD takes ceil(units / 4) units of ITS tower; else 8 XCDs share the units
        const long slots = e.tower_split ? (units + 3) / 4 : (units + 7) / 8;
        e.pair_bf16 = (e.tower_split && e.round_bf16 && !getenv("VAA_K2E_F32_PAIRS")) ? 1 : 0;
        const dim3 grid((unsigned)(slots * 8 * nch)), blk(kEmbedFastThreads);
        if (!e.tower_split) VAA_LAUNCH((embed_dgrad_tiles_lds_kernel<false, 2>), grid, blk, lds_fast, st, e, nch);
        else if (nbmax == 1) VAA_LAUNCH((embed_dgrad_tiles_lds_kernel<true, 1>), grid, blk, lds_fast, st, e, nch);
        else if (nbmax == 2) VAA_LAUNCH((embed_dgrad_tiles_lds_kernel<true, 2>), grid, blk, lds_fast, st, e, nch);
        else VAA_LAUNCH((embed_dgrad_tiles_lds_kernel<true, 3>), grid, blk, lds_fast, st, e, nch);
    } else {  // wide towers: fragments straight from global memory
        const int nch = (kNBlocks + 3) / 4;  // 10 workgroups per image
        VAA_LAUNCH(embed_dgrad_tiles_kernel, dim3((unsigned)((B + 7) / 8 * 8) * nch, ny), dim3(kEmbedThreads), 0, st, e, nch);
    }
    return VAA_OK;
}

}  // namespace vaa

extern "C" size_t vaa_patch_embed_packed_elems(int D) {
    if (D <= 0 || (D % 64) != 0) return 0;
    return (size_t)vaa::kNBlocks * 16 * (size_t)D;  // 592 columns (588 + 4 of zero padding) x D
}

extern "C" int vaa_patch_embed_pack_weights(const uint16_t* wt, int D, uint16_t* packed, void* stream) {
    using namespace vaa;
    if (!wt || !packed) {
        set_error("vaa_patch_embed_pack_weights: null pointer argument");
        return VAA_E_INVALID;
    }
    if (D <= 0 || (D % 64) != 0) {
        set_error("vaa_patch_embed_pack_weights: tower width %d is not a positive multiple of 64", D);
        return VAA_E_INVALID;
    }
    const long nfrag = (long)kNBlocks * (D >> 6) * 2 * 64;
    VAA_LAUNCH(embed_pack_weights_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wt, D, packed);
    return check_launch("vaa_patch_embed_pack_weights");
}

extern "C" size_t vaa_patch_embed_grad_ws_bytes(int B, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0) return 0;
    return vaa_patch_grad_ws_bytes(B, ph, pw) + 2 * (size_t)B * 256 * vaa::kTileElems * sizeof(float) + 256;  // partial tiles + one tile-gradient buffer per tower
}

namespace vaa {

// K2' for one patch per batch. keep_tiles / tile_flags (K1's tile-major outputs) replace keep_bits when given; gpatch == nullptr leaves the
// fixed-order sum of the partial tiles (ws[0 .. parts*3*ph*pw), parts = vaa_patch_grad_partials(B)) to the caller's vaa_step_epilogue.
static int embed_grad_gather_impl(const char* who, const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wt0, const uint16_t* wt1,
                                  const float* patch, const int32_t* xy, const float* theta, const uint8_t* keep_bits, const uint16_t* keep_tiles,
                                  const uint32_t* tile_flags, int B, int ph, int pw, int geometry, int mask_mode, const float* std6,
                                  int round_bf16, float* gpatch, bool defer_reduce, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 && gpatch && ph > 0 && pw > 0) {
        if (hipMemsetAsync(gpatch, 0, (size_t)3 * ph * pw * sizeof(float), st) != hipSuccess) return check_launch(who);
        return VAA_OK;
    }
    const bool tiled_keep = keep_tiles != nullptr;
    if (!dy0 || !dy1 || !wt0 || !wt1 || !xy || !std6 || (!gpatch && !defer_reduce) || (!keep_bits && !tiled_keep) || (tiled_keep && !tile_flags) ||
        (geometry && !theta)) {
        set_error("%s: null pointer argument (the keep mask of K1 is required)", who);
        return VAA_E_INVALID;
    }
    if (B <= 0 || ph <= 0 || pw <= 0 || D0 <= 0 || D1 <= 0 || (D0 % 64) != 0 || (D1 % 64) != 0 ||
        (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("%s: bad sizes/mode (B=%d ph=%d pw=%d D0=%d D1=%d; D %% 64 == 0)", who, B, ph, pw, D0, D1);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("%s: patch %dx%d larger than the frame", who, ph, pw);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        set_error("%s: VAA_MASK_NE_M100 is defined for geometry=0 only", who);
        return VAA_E_UNSUPPORTED;
    }
    if (!ws || ws_bytes < vaa_patch_embed_grad_ws_bytes(B, ph, pw)) {
        set_error("%s: workspace %zu B < required %zu B", who, ws_bytes, vaa_patch_embed_grad_ws_bytes(B, ph, pw));
        return VAA_E_WORKSPACE;
    }
    char* wsb = reinterpret_cast<char*>(ws);
    const size_t part_bytes = (vaa_patch_grad_ws_bytes(B, ph, pw) + 255) / 256 * 256;
    EmbedArgs e;
    e.dy0 = dy0; e.dy1 = dy1; e.wt0 = wt0; e.wt1 = wt1; e.keep = keep_bits; e.flags = tile_flags;
    e.geff = reinterpret_cast<float*>(wsb + part_bytes);
    e.geff2 = e.geff + (size_t)B * 256 * kTileElems;
    e.B = B; e.D0 = D0; e.D1 = D1; e.round_bf16 = round_bf16 ? 1 : 0;
    for (int q = 0; q < 6; ++q) e.istd6[q] = (float)(1.0 / (double)std6[q]);
    if (launch_embed_tiles(e, ph, pw, st, who) != VAA_OK) return VAA_E_LAUNCH;
    int rc = check_launch(who);
    if (rc != VAA_OK) return rc;
    GradArgs a;
    a.g = nullptr; a.patch = patch; a.xy = xy; a.theta = theta; a.keep = keep_bits; a.partial = (float*)ws; a.pdesc = nullptr;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) a.istd6[q] = e.istd6[q];
    a.geff = e.geff; a.geff2 = e.tower_split ? e.geff2 : nullptr; a.geff_bf16 = e.pair_bf16; a.keep_t = keep_tiles;
    if (keep_tiles) a.keep = reinterpret_cast<const uint8_t*>(keep_tiles);  // non-null selects the stored-mask instantiation
    return launch_scatter_reduce<true>(a, defer_reduce ? nullptr : gpatch, st, who);
}

}  // namespace vaa

extern "C" int vaa_patch_embed_grad_gather(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wt0, const uint16_t* wt1,
                                           const float* patch, const int32_t* xy, const float* theta, const uint8_t* keep_bits, int B, int ph,
                                           int pw, int geometry, int mask_mode, const float* std6, int round_bf16, float* gpatch, void* ws,
                                           size_t ws_bytes, void* stream) {
    return vaa::embed_grad_gather_impl("vaa_patch_embed_grad_gather", dy0, D0, dy1, D1, wt0, wt1, patch, xy, theta, keep_bits, nullptr, nullptr, B, ph,
                                       pw, geometry, mask_mode, std6, round_bf16, gpatch, false, ws, ws_bytes, stream);
}

extern "C" int vaa_patch_grad_partials(int B) { return B > 0 ? vaa::grad_sched(B).gx : 0; }

extern "C" int vaa_patch_embed_grad_gather_tiles(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wt0, const uint16_t* wt1,
                                                 const float* patch, const int32_t* xy, const float* theta, const uint16_t* keep_tiles,
                                                 const uint32_t* tile_flags, int B, int ph, int pw, int geometry, int mask_mode, const float* std6,
                                                 int round_bf16, float* gpatch, void* ws, size_t ws_bytes, void* stream) {
    return vaa::embed_grad_gather_impl("vaa_patch_embed_grad_gather_tiles", dy0, D0, dy1, D1, wt0, wt1, patch, xy, theta, nullptr, keep_tiles, tile_flags,
                                       B, ph, pw, geometry, mask_mode, std6, round_bf16, gpatch, gpatch == nullptr, ws, ws_bytes, stream);
}

extern "C" size_t vaa_patch_embed_grad_multi_ws_bytes(int B) {
    if (B <= 0) return 0;
    return 2 * (size_t)B * 256 * vaa::kTileElems * sizeof(float) + 256;
}

namespace vaa {

// K2' with one patch per image (resize_patch=True): the tile gradients do not depend on the patches, the gather runs in MULTI mode.
static int embed_grad_gather_multi_impl(const char* who, const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                        const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta, const uint8_t* keep_bits,
                                        const uint16_t* keep_tiles, const uint32_t* tile_flags, int B, int max_h, int max_w, int geometry, int mask_mode,
                                        const float* std6, int round_bf16, float* gpacked, void* ws, size_t ws_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return VAA_OK;
    if (!dy0 || !dy1 || !wp0 || !wp1 || !pdesc || !xy || !std6 || !gpacked || (!keep_bits && !keep_tiles) || (keep_tiles && !tile_flags) || (geometry && !theta)) {
        set_error("%s: null pointer argument (the keep mask of K1 is required)", who);
        return VAA_E_INVALID;
    }
    if (B < 0 || max_h <= 0 || max_w <= 0 || D0 <= 0 || D1 <= 0 || (D0 % 64) != 0 || (D1 % 64) != 0 ||
        (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("%s: bad sizes/mode (B=%d max_h=%d max_w=%d D0=%d D1=%d; D %% 64 == 0)", who, B, max_h, max_w, D0, D1);
        return VAA_E_INVALID;
    }
    if (max_h > VAA_IMG || max_w > VAA_IMG) {
        set_error("%s: patch bound %dx%d larger than the frame", who, max_h, max_w);
        return VAA_E_UNSUPPORTED;
    }
    if (geometry && mask_mode == VAA_MASK_NE_M100) {
        set_error("%s: VAA_MASK_NE_M100 is defined for geometry=0 only", who);
        return VAA_E_UNSUPPORTED;
    }
    if (!ws || ws_bytes < vaa_patch_embed_grad_multi_ws_bytes(B)) {
        set_error("%s: workspace %zu B < required %zu B", who, ws_bytes, vaa_patch_embed_grad_multi_ws_bytes(B));
        return VAA_E_WORKSPACE;
    }
    EmbedArgs e;
    e.dy0 = dy0; e.dy1 = dy1; e.wt0 = wp0; e.wt1 = wp1; e.keep = keep_bits; e.flags = tile_flags; e.geff = reinterpret_cast<float*>(ws);
    e.geff2 = e.geff + (size_t)B * 256 * kTileElems;
    e.B = B; e.D0 = D0; e.D1 = D1; e.round_bf16 = round_bf16 ? 1 : 0;
    for (int q = 0; q < 6; ++q) e.istd6[q] = (float)(1.0 / (double)std6[q]);
    if (launch_embed_tiles(e, max_h, max_w, st, who) != VAA_OK) return VAA_E_LAUNCH;
    int rc = check_launch(who);
    if (rc != VAA_OK) return rc;
    GradArgs a;
    a.g = nullptr; a.patch = packed; a.xy = xy; a.theta = theta; a.keep = keep_bits; a.partial = gpacked; a.pdesc = pdesc;
    a.B = B; a.ph = max_h; a.pw = max_w; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) a.istd6[q] = e.istd6[q];
    a.geff = e.geff; a.geff2 = e.tower_split ? e.geff2 : nullptr; a.geff_bf16 = e.pair_bf16; a.keep_t = keep_tiles;
    if (keep_tiles) a.keep = reinterpret_cast<const uint8_t*>(keep_tiles);  // non-null selects the stored-mask instantiation
    return launch_scatter_multi<true>(a, max_h, max_w, st, who);
}

}  // namespace vaa

extern "C" int vaa_patch_embed_grad_gather_multi(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                                 const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                                 const uint8_t* keep_bits, int B, int max_h, int max_w, int geometry, int mask_mode,
                                                 const float* std6, int round_bf16, float* gpacked, void* ws, size_t ws_bytes, void* stream) {
    return vaa::embed_grad_gather_multi_impl("vaa_patch_embed_grad_gather_multi", dy0, D0, dy1, D1, wp0, wp1, packed, pdesc, xy, theta, keep_bits, nullptr, nullptr,
                                             B, max_h, max_w, geometry, mask_mode, std6, round_bf16, gpacked, ws, ws_bytes, stream);
}

extern "C" int vaa_patch_embed_grad_gather_multi_tiles(const uint16_t* dy0, int D0, const uint16_t* dy1, int D1, const uint16_t* wp0, const uint16_t* wp1,
                                                       const float* packed, const int32_t* pdesc, const int32_t* xy, const float* theta,
                                                       const uint16_t* keep_tiles, const uint32_t* tile_flags, int B, int max_h, int max_w, int geometry,
                                                       int mask_mode, const float* std6, int round_bf16, float* gpacked, void* ws, size_t ws_bytes,
                                                       void* stream) {
    return vaa::embed_grad_gather_multi_impl("vaa_patch_embed_grad_gather_multi_tiles", dy0, D0, dy1, D1, wp0, wp1, packed, pdesc, xy, theta, nullptr, keep_tiles,
                                             tile_flags, B, max_h, max_w, geometry, mask_mode, std6, round_bf16, gpacked, ws, ws_bytes, stream);
}
