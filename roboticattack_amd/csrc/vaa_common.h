// vaa_common.h — shared device helpers for the gfx950 kernels behind include/vaa.h.
//
// Numerics contract (re-verified against the reference in tests/golden):
// this translation unit is compiled with -ffp-contract=off; every fused multiply-add below is
// explicit so the warp coordinates, bilinear weights and the `canvas < -20` mask reproduce the
// reference's PyTorch-CPU path (F.affine_grid + F.grid_sample, appply_random_transform.py:93-102)
// bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/vaa.h"

#define VAA_NPIX (VAA_IMG * VAA_IMG)

namespace vaa {

void set_error(const char* fmt, ...);
int check_launch(const char* what);  // the HIP runtime's last error, then async_error_poll()
unsigned* async_error_word();        // pinned, device-mapped word kernels OR failure bits into (nullptr: allocation failed)
int async_error_poll(const char* what, bool clear);
constexpr unsigned VAA_ASYNC_K3_HANDOVER_TIMEOUT = 1u;
// vaa_patch_grad.hip: gpatch[e] = sum_p partial[p][e] (p < nparts, e < n) in a fixed order with fp64 accumulation
int launch_partial_reduce(const float* partial, float* gpatch, int n, int nparts, hipStream_t st, const char* who);

// Per-dispatch timing (include/vaa.h: vaa_prof_*): while the profiler is armed every kernel of the library is dispatched through
// hipExtLaunchKernel with its own start/stop event pair, which the runtime binds to THAT dispatch's begin/end timestamps (the figures
// rocprofv3 --kernel-trace reports), so a kernel's duration inside a real step is read without bracketing markers.
bool prof_next(const char* name, hipEvent_t* start, hipEvent_t* stop);

template <typename F, typename... Args>
inline void launch_k(const char* name, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipEvent_t s, e;
    if (prof_next(name, &s, &e)) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, s, e, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}
#define VAA_LAUNCH(kernel, grid, block, lds, stream, ...) ::vaa::launch_k(#kernel, kernel, grid, block, lds, stream, __VA_ARGS__)

struct Norm6 {
    float mean[6];
    float stdv[6];
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// torch `.to(torch.bfloat16)`: round-to-nearest-even; NaN stays NaN.
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// F.affine_grid(align_corners=False) base coordinate of pixel index i along a 224-long axis:
// torch.linspace(-1, 1, 224)[i] * 223 / 224, with linspace's two-sided evaluation (one FMA each side).
__device__ __forceinline__ float base_coord(int i) {
    const float step = 2.0f / 223.0f;
    float lin = (i < VAA_IMG / 2) ? __builtin_fmaf((float)i, step, -1.0f)
                                  : __builtin_fmaf(-(float)(VAA_IMG - 1 - i), step, 1.0f);
    return (lin * 223.0f) / 224.0f;
}

struct Samp {
    int x0, y0;            // floor of the clamped source position (north-west corner)
    float nw, ne, sw, se;  // bilinear corner weights
};

// bx/by: base coords of output column j / row i; th: row-major 2x3 affine.
// grid = base @ theta^T as torch's CPU GEMM evaluates it (product, one FMA, plain add), then
// grid_sample's unnormalise contracted into one FMA, clamp to the border, floor, weights.
__device__ __forceinline__ Samp sample_pos(float bx, float by, const float* th) {
    Samp s;
    float gx = __builtin_fmaf(by, th[1], bx * th[0]) + th[2];
    float gy = __builtin_fmaf(by, th[4], bx * th[3]) + th[5];
    float ix = __builtin_fmaf(gx + 1.0f, 112.0f, -0.5f);
    float iy = __builtin_fmaf(gy + 1.0f, 112.0f, -0.5f);
    ix = fminf(223.0f, fmaxf(ix, 0.0f));
    iy = fminf(223.0f, fmaxf(iy, 0.0f));
    float xw = floorf(ix), yn = floorf(iy);
    float w = ix - xw, e = 1.0f - w, n = iy - yn, so = 1.0f - n;
    s.x0 = (int)xw;
    s.y0 = (int)yn;
    s.nw = so * e;
    s.ne = so * w;
    s.sw = n * e;
    s.se = n * w;
    return s;
}

// The same coordinates with the fractional parts kept (K2 stores {x0, y0, w, n} per pixel and forms the four corner weights
// when it scatters: so*e, so*w, n*e, n*w with e = 1-w, so = 1-n — the products of sample_pos above, bit for bit).
__device__ __forceinline__ void sample_pos_frac(float bx, float by, const float* th, int& x0, int& y0, float& w, float& n) {
    float gx = __builtin_fmaf(by, th[1], bx * th[0]) + th[2];
    float gy = __builtin_fmaf(by, th[4], bx * th[3]) + th[5];
    float ix = __builtin_fmaf(gx + 1.0f, 112.0f, -0.5f);
    float iy = __builtin_fmaf(gy + 1.0f, 112.0f, -0.5f);
    ix = fminf(223.0f, fmaxf(ix, 0.0f));
    iy = fminf(223.0f, fmaxf(iy, 0.0f));
    float xw = floorf(ix), yn = floorf(iy);
    w = ix - xw;
    n = iy - yn;
    x0 = (int)xw;
    y0 = (int)yn;
}

__device__ __forceinline__ Samp samp_from_frac(int x0, int y0, float w, float n) {
    Samp s;
    const float e = 1.0f - w, so = 1.0f - n;
    s.x0 = x0; s.y0 = y0;
    s.nw = so * e; s.ne = so * w; s.sw = n * e; s.se = n * w;
    return s;
}

// canvas = -100 everywhere with the patch pasted at (px,py) (appply_random_transform.py:111,125);
// corners beyond the frame contribute 0 (grid_sample's bounds mask).
template <typename P>
__device__ __forceinline__ float canvas_at(P patch_c, int ph, int pw, int px, int py, int xx, int yy) {
    if (xx >= VAA_IMG || yy >= VAA_IMG) return 0.0f;
    int u = xx - px, v = yy - py;
    if ((unsigned)u < (unsigned)pw && (unsigned)v < (unsigned)ph) return patch_c[v * pw + u];
    return -100.0f;
}

template <typename P>
__device__ __forceinline__ float sample_canvas(P patch_c, int ph, int pw, int px, int py, const Samp& s) {
    float vnw = canvas_at(patch_c, ph, pw, px, py, s.x0, s.y0);
    float vne = canvas_at(patch_c, ph, pw, px, py, s.x0 + 1, s.y0);
    float vsw = canvas_at(patch_c, ph, pw, px, py, s.x0, s.y0 + 1);
    float vse = canvas_at(patch_c, ph, pw, px, py, s.x0 + 1, s.y0 + 1);
    return __builtin_fmaf(vse, s.se, __builtin_fmaf(vsw, s.sw, __builtin_fmaf(vne, s.ne, vnw * s.nw)));
}

__device__ __forceinline__ bool keep_rule(float cv, int mask_mode) {
    return mask_mode == VAA_MASK_LT_M20 ? !(cv < -20.0f) : (cv != -100.0f);
}

// Pixel-space view of the same affine map (used only to BOUND the footprint, never for values):
//   ix ~= a00*j + a01*i + c0,  iy ~= a10*j + a11*i + c1.
struct PixAffine {
    float a00, a01, c0, a10, a11, c1;
};

__device__ __forceinline__ PixAffine pix_affine(const float* th) {
    PixAffine p;
    p.a00 = th[0];
    p.a01 = th[1];
    p.c0 = 112.0f * (th[2] + 1.0f) - 0.5f - 111.5f * (th[0] + th[1]);
    p.a10 = th[3];
    p.a11 = th[4];
    p.c1 = 112.0f * (th[5] + 1.0f) - 0.5f - 111.5f * (th[3] + th[4]);
    return p;
}

// j-interval of row i whose (approximate, unclamped) source coordinate a*j + base lies in [lo, hi)
__device__ __forceinline__ void solve_interval(float a, float base, float lo, float hi, float& jl, float& jh) {
    if (fabsf(a) > 1e-6f) {
        float t0 = (lo - base) / a, t1 = (hi - base) / a;
        jl = fmaxf(jl, fminf(t0, t1));
        jh = fminf(jh, fmaxf(t0, t1));
    } else if (!(base >= lo - 1.0f && base < hi + 1.0f)) {
        jl = 1e30f;
        jh = -1e30f;
    }
}

// K4's per-element arithmetic (vaa_update.hip; also applied by the step epilogue when the update is fused into it: one source, same bits)
struct UpdArgs {
    float* patch;
    const float* g;
    float* m;
    float* v;
    float* stats;
    int n, mode;
    float lr, b1, b2, eps, step_size, one_m_b1, one_m_b2, l1_clip, grad_scale;
};

__device__ __forceinline__ float update_one(const UpdArgs& a, float g, float p, float& m, float& v) {
    if (a.mode == VAA_OPT_ADAMW_HF) {
        m = __builtin_fmaf(g, a.one_m_b1, m * a.b1);      // exp_avg.mul_(b1).add_(g, alpha=1-b1)
        v = v * a.b2 + (a.one_m_b2 * g) * g;              // exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)
        const float denom = sqrtf(v) + a.eps;             // v.sqrt().add_(eps)
        p = p + ((-a.step_size) * m) / denom;             // p.addcdiv_(m, denom, value=-step_size)
    } else {
        const float sg = (g > 0.0f) ? 1.0f : ((g < 0.0f) ? -1.0f : 0.0f);
        p = p - a.lr * sg;
    }
    return fminf(1.0f, fmaxf(0.0f, p));                   // patch.data.clamp(0, 1)
}

// out[e] = sum_p partial[p][e] (p < nparts, e < n) for the 64 elements of block `blk`, in a fixed two-level order (16 interleaved slices,
// then slice 0..15), fp64: the body of patch_grad_reduce_kernel, shared with the step epilogue (256 threads; sl = 16 KB of LDS).
// A thread owns four consecutive elements (16 B loads); block = 16 element-quads x 16 slices.
// Returns true in the threads that own a result element (index `oe`, value `ov`) after storing it.
__device__ __forceinline__ bool partial_reduce_block(const float* __restrict__ partial, float* __restrict__ out, int n, int nparts, int blk,
                                                     double (*sl)[16][4], int& oe, float& ov) {
    const int el = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int e = (blk * 16 + el) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (e + 3 < n && (n & 3) == 0) {
#pragma unroll 8
        for (int p = s; p < nparts; p += 16) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)p * n + e);
            acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
        }
    } else {
        for (int p = s; p < nparts; p += 16)
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (e + z < n) acc[z] += (double)partial[(size_t)p * n + e + z];
    }
#pragma unroll
    for (int z = 0; z < 4; ++z) sl[s][el][z] = acc[z];
    __syncthreads();
    if (s < 4) {  // thread (s, el) of the second level sums element z = s of quad el
        const int z = s;
        if (e + z < n) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += sl[q][el][z];
            out[e + z] = (float)t;
            oe = e + z;
            ov = (float)t;
            return true;
        }
    }
    return false;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace vaa
