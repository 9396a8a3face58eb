// vaa_model_ops.hip — OPTIONAL fused elementwise operators for the PyTorch-ROCm model around the hot path.
//
// NOT part of the SURVEY.md section-8 contract: the reference keeps the OpenVLA forward/backward as stock PyTorch and so
// does this build (roboticattack_amd/openvla_model.py). rocprofv3 shows 25 % of the bs=64 step in eager elementwise /
// copy kernels of the Llama layers (RoPE = neg + cat + 2 mul + add per tensor, SwiGLU = silu + mul and their multi-kernel
// backwards); the two kernels below replace those chains by single HBM-bound passes. The model falls back to the eager
// PyTorch ops when they are disabled (VAA_NO_FUSED_MODEL_OPS=1) or the tensors are not bf16 on a ROCm device.
#include <stdlib.h>

#include "vaa_common.h"

namespace vaa {

// ---- rotary embedding, forward and backward (the backward is the same rotation with -sin) ----
// x: bf16 [B,T,H,hd] addressed through element strides (sb, st, sh), last dim contiguous; out: bf16 contiguous [B,T,H,hd].
// out[..., i]        = x[i]*cos[t,i]        - x[i+hd/2]*sin[t,i]*sgn
// out[..., i+hd/2]   = x[i+hd/2]*cos[t,i]   + x[i]*sin[t,i]*sgn                 (HF rotate_half convention)
struct RopeArgs {
    const uint16_t* x;
    uint16_t* out;
    const float* cos;  // [T, hd/2]
    const float* sin;  // [T, hd/2]
    long sb, st, sh;
    int B, T, H, hd;
    float sgn;
};

__global__ __launch_bounds__(256) void rope_kernel(const RopeArgs a) {
    const int half = a.hd >> 1, vec_per_row = half >> 3;  // 8 bf16 per 16-byte vector
    const long nvec = (long)a.B * a.T * a.H * vec_per_row;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        const int iv = (int)(v % vec_per_row);
        const long row = v / vec_per_row;  // (b*T + t)*H + h
        const int h = (int)(row % a.H);
        const long bt = row / a.H;
        const int t = (int)(bt % a.T), b = (int)(bt / a.T);
        const uint16_t* xp = a.x + (long)b * a.sb + (long)t * a.st + (long)h * a.sh + iv * 8;
        const uint4 lo = *reinterpret_cast<const uint4*>(xp), hi = *reinterpret_cast<const uint4*>(xp + half);
        const float4 c0 = *reinterpret_cast<const float4*>(a.cos + (long)t * half + iv * 8), c1 = *reinterpret_cast<const float4*>(a.cos + (long)t * half + iv * 8 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(a.sin + (long)t * half + iv * 8), s1 = *reinterpret_cast<const float4*>(a.sin + (long)t * half + iv * 8 + 4);
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t lw[4] = {lo.x, lo.y, lo.z, lo.w}, hw[4] = {hi.x, hi.y, hi.z, hi.w};
        uint32_t ol[4], oh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x1a = __uint_as_float(lw[q] << 16), x1b = __uint_as_float(lw[q] & 0xffff0000u);
            const float x2a = __uint_as_float(hw[q] << 16), x2b = __uint_as_float(hw[q] & 0xffff0000u);
            const float sa = sn[2 * q] * a.sgn, sb2 = sn[2 * q + 1] * a.sgn;
            const float o1a = x1a * cs[2 * q] - x2a * sa, o1b = x1b * cs[2 * q + 1] - x2b * sb2;
            const float o2a = x2a * cs[2 * q] + x1a * sa, o2b = x2b * cs[2 * q + 1] + x1b * sb2;
            ol[q] = f32_to_bf16_bits(o1a) | (f32_to_bf16_bits(o1b) << 16);
            oh[q] = f32_to_bf16_bits(o2a) | (f32_to_bf16_bits(o2b) << 16);
        }
        uint16_t* op = a.out + row * a.hd + iv * 8;
        *reinterpret_cast<uint4*>(op) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
        *reinterpret_cast<uint4*>(op + half) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
    }
}

// ---- SwiGLU: y = silu(g) * u ;  backward: dg = dy * u * sig(g) * (1 + g*(1 - sig(g))), du = dy * silu(g) ----
__device__ __forceinline__ void unpack8(const uint4& r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16), f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16),
                      f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16), f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16));
}

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ u, uint16_t* __restrict__ y, long nvec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float gv[8], uv[8], o[8];
        unpack8(reinterpret_cast<const uint4*>(g)[v], gv);
        unpack8(reinterpret_cast<const uint4*>(u)[v], uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gv[e] / (1.0f + __expf(-gv[e])) * uv[e];
        reinterpret_cast<uint4*>(y)[v] = pack8(o);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ g, const uint16_t* __restrict__ u,
                                                          uint16_t* __restrict__ dg, uint16_t* __restrict__ du, long nvec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float dv[8], gv[8], uv[8], og[8], ou[8];
        unpack8(reinterpret_cast<const uint4*>(dy)[v], dv);
        unpack8(reinterpret_cast<const uint4*>(g)[v], gv);
        unpack8(reinterpret_cast<const uint4*>(u)[v], uv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = 1.0f / (1.0f + __expf(-gv[e]));
            ou[e] = dv[e] * gv[e] * sg;
            og[e] = dv[e] * uv[e] * sg * (1.0f + gv[e] * (1.0f - sg));
        }
        reinterpret_cast<uint4*>(dg)[v] = pack8(og);
        reinterpret_cast<uint4*>(du)[v] = pack8(ou);
    }
}


// ---- LayerScale + residual of the DINOv2 blocks: out = x + a * ls (ls bf16 [D] broadcast over the rows; x == nullptr: out = a * ls, the
//      backward's d a = g * ls). torch.addcmul with the broadcast operand runs its strided fallback at 0.8 TB/s (128 us for three 34 MB tensors). ----
__global__ __launch_bounds__(256) void scale_add_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ a, const uint16_t* __restrict__ ls,
                                                         uint16_t* __restrict__ out, long nvec, int dvec) {
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long)gridDim.x * 256) {
        float av[8], lv[8], xv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
        unpack8(reinterpret_cast<const uint4*>(a)[v], av);
        unpack8(reinterpret_cast<const uint4*>(ls)[v % dvec], lv);
        if (x) unpack8(reinterpret_cast<const uint4*>(x)[v], xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(av[e], lv[e], xv[e]);
        reinterpret_cast<uint4*>(out)[v] = pack8(o);
    }
}

// ---- RMSNorm with residual pass-through: h = x * rstd * w ; backward: gx = g_pass + rstd*(gh*w - xhat*mean(gh*w*xhat)) ----
// One 256-thread workgroup per row of D bf16 (D % 8 == 0, D <= 8192); the row lives in registers between the two passes.
constexpr int kNormThreads = 256, kNormMaxVec = 4;  // 4 x 8 elements per thread -> D <= 8192

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(kNormThreads) void rmsnorm_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, uint16_t* __restrict__ h,
                                                                    float* __restrict__ rstd, int D, float eps) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const int nv = D >> 3;
    float xv[kNormMaxVec][8];
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xv[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += xv[c][e] * xv[c][e];
        }
    }
    ss = block_sum_256(ss, sh);
    const float r = rsqrtf(ss / (float)D + eps);
    if (threadIdx.x == 0) rstd[row] = r;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float wv[8], o[8];
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = bf16_bits_to_f32(f32_to_bf16_bits(xv[c][e] * r)) * wv[e];  // HF: (x*rstd).to(bf16) * weight
            reinterpret_cast<uint4*>(h + row * D)[q] = pack8(o);
        }
    }
}

__global__ __launch_bounds__(kNormThreads) void rmsnorm_bwd_kernel(const uint16_t* __restrict__ gh, const uint16_t* __restrict__ gpass,
                                                                    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                    const float* __restrict__ rstd, uint16_t* __restrict__ gx, int D) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const int nv = D >> 3;
    const float r = rstd[row];
    float xh[kNormMaxVec][8], gw[kNormMaxVec][8];
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float wv[8];
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xh[c]);
            unpack8(reinterpret_cast<const uint4*>(gh + row * D)[q], gw[c]);
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[c][e] *= r;
                gw[c][e] *= wv[e];
                dot += gw[c][e] * xh[c][e];
            }
        }
    }
    dot = block_sum_256(dot, sh) / (float)D;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float o[8], gp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gpass) unpack8(reinterpret_cast<const uint4*>(gpass + row * D)[q], gp);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gp[e] + r * (gw[c][e] - xh[c][e] * dot);
            reinterpret_cast<uint4*>(gx + row * D)[q] = pack8(o);
        }
    }
}

// ---- LayerNorm with residual pass-through (ViT blocks): h = (x - mean) * rstd * w + b, statistics in fp32, one rounding;
//      backward: gx = g_pass + rstd * (gh*w - mean(gh*w) - xhat * mean(gh*w*xhat)). w, b frozen (no gradients). ----
__device__ __forceinline__ void block_sum2_256(float& a, float& b, float (*sh)[2]) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = a; sh[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    a = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
    b = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
}

__global__ __launch_bounds__(kNormThreads) void layernorm_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                      const uint16_t* __restrict__ b, uint16_t* __restrict__ h,
                                                                      float* __restrict__ stats, int D, float eps) {
    __shared__ float sh[4];
    const long row = blockIdx.x;
    const int nv = D >> 3;
    float xv[kNormMaxVec][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xv[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xv[c][e];
        }
    }
    const float mean = block_sum_256(s, sh) / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[c][e] -= mean; ss += xv[c][e] * xv[c][e]; }
        }
    }
    const float r = rsqrtf(block_sum_256(ss, sh) / (float)D + eps);
    if (threadIdx.x == 0) { stats[2 * row] = mean; stats[2 * row + 1] = r; }
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float wv[8], bv[8], o[8];
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
            unpack8(reinterpret_cast<const uint4*>(b)[q], bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = xv[c][e] * r * wv[e] + bv[e];
            reinterpret_cast<uint4*>(h + row * D)[q] = pack8(o);
        }
    }
}

__global__ __launch_bounds__(kNormThreads) void layernorm_bwd_kernel(const uint16_t* __restrict__ gh, const uint16_t* __restrict__ gpass,
                                                                      const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                      const float* __restrict__ stats, uint16_t* __restrict__ gx, int D) {
    __shared__ float sh2[4][2];
    const long row = blockIdx.x;
    const int nv = D >> 3;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    float xh[kNormMaxVec][8], gw[kNormMaxVec][8];
    float sg = 0.0f, dot = 0.0f;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float wv[8];
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xh[c]);
            unpack8(reinterpret_cast<const uint4*>(gh + row * D)[q], gw[c]);
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[c][e] = (xh[c][e] - mean) * r;
                gw[c][e] *= wv[e];
                sg += gw[c][e];
                dot += gw[c][e] * xh[c][e];
            }
        }
    }
    block_sum2_256(sg, dot, sh2);
    sg /= (float)D;
    dot /= (float)D;
#pragma unroll
    for (int c = 0; c < kNormMaxVec; ++c) {
        const int q = threadIdx.x + c * kNormThreads;
        if (q < nv) {
            float o[8], gp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gpass) unpack8(reinterpret_cast<const uint4*>(gpass + row * D)[q], gp);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gp[e] + r * (gw[c][e] - sg - xh[c][e] * dot);
            reinterpret_cast<uint4*>(gx + row * D)[q] = pack8(o);
        }
    }
}

// ---- the same two operators for NARROW rows (the ViT towers: D = 1024 / 1152, 2 KB rows): ONE WAVE per row, four rows per workgroup, no
//      LDS and no workgroup barrier — the row's sums are wave reductions. One 256-thread workgroup per 2 KB row left half its threads idle at
//      D = 1024 and paid two (fwd: four) barriers per row: 75 us forward / 98 us backward for 16,448 rows at bs=64 = 0.9-1.0 TB/s, the slowest
//      streams of the step (profiles/r04_bench_kernel_stats.csv). NV = 16-byte vectors per lane (D <= NV * 512). Same arithmetic per element;
//      only the order of the fp32 row sums differs. ----
constexpr int kNormWaveRows = 4;

template <int NV>
__global__ __launch_bounds__(64 * kNormWaveRows) void layernorm_fwd_wave_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                                 const uint16_t* __restrict__ b, uint16_t* __restrict__ h,
                                                                                 float* __restrict__ stats, long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * kNormWaveRows + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int nv = D >> 3;
    float xv[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int q = lane + c * 64;
        if (q < nv) {
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xv[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xv[c][e];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        if (lane + c * 64 < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[c][e] -= mean; ss += xv[c][e] * xv[c][e]; }
        }
    }
    const float r = rsqrtf(wave_sum(ss) / (float)D + eps);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = r; }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int q = lane + c * 64;
        if (q < nv) {
            float wv[8], bv[8], o[8];
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
            unpack8(reinterpret_cast<const uint4*>(b)[q], bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = xv[c][e] * r * wv[e] + bv[e];
            reinterpret_cast<uint4*>(h + row * D)[q] = pack8(o);
        }
    }
}

template <int NV>
__global__ __launch_bounds__(64 * kNormWaveRows) void layernorm_bwd_wave_kernel(const uint16_t* __restrict__ gh, const uint16_t* __restrict__ gpass,
                                                                                 const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                                                 const float* __restrict__ stats, uint16_t* __restrict__ gx, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * kNormWaveRows + (threadIdx.x >> 6);
    if (row >= rows) return;  // wave-uniform
    const int nv = D >> 3;
    const float mean = stats[2 * row], r = stats[2 * row + 1];
    float xh[NV][8], gw[NV][8], gp[NV][8];
    float sg = 0.0f, dot = 0.0f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int q = lane + c * 64;
        if (q < nv) {  // all of the row's loads (x, gh, g_pass) in flight before the first use
            float wv[8];
            unpack8(reinterpret_cast<const uint4*>(x + row * D)[q], xh[c]);
            unpack8(reinterpret_cast<const uint4*>(gh + row * D)[q], gw[c]);
            if (gpass) unpack8(reinterpret_cast<const uint4*>(gpass + row * D)[q], gp[c]);
            unpack8(reinterpret_cast<const uint4*>(w)[q], wv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[c][e] = (xh[c][e] - mean) * r;
                gw[c][e] *= wv[e];
                sg += gw[c][e];
                dot += gw[c][e] * xh[c][e];
            }
        }
    }
    sg = wave_sum(sg) / (float)D;
    dot = wave_sum(dot) / (float)D;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int q = lane + c * 64;
        if (q < nv) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (gpass ? gp[c][e] : 0.0f) + r * (gw[c][e] - sg - xh[c][e] * dot);
            reinterpret_cast<uint4*>(gx + row * D)[q] = pack8(o);
        }
    }
}

static bool ln_wave_enabled() {  // VAA_LN_WAVE=0: the one-workgroup-per-row kernels for every width (A/B)
    static const bool on = [] { const char* e = getenv("VAA_LN_WAVE"); return !(e && e[0] == '0'); }();
    return on;
}

static unsigned stream_grid(long nvec) {
    long b = (nvec + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace vaa

extern "C" int vaa_model_rope(const uint16_t* x, long sb, long st, long sh, const float* cos_t, const float* sin_t, int B, int T, int H, int hd,
                              float sin_sign, uint16_t* out, void* stream) {
    using namespace vaa;
    if (!x || !cos_t || !sin_t || !out) { set_error("vaa_model_rope: null pointer argument"); return VAA_E_INVALID; }
    if (B <= 0 || T <= 0 || H <= 0 || hd <= 0 || (hd % 16) != 0) { set_error("vaa_model_rope: bad sizes (hd must be a multiple of 16)"); return VAA_E_INVALID; }
    RopeArgs a;
    a.x = x; a.out = out; a.cos = cos_t; a.sin = sin_t; a.sb = sb; a.st = st; a.sh = sh; a.B = B; a.T = T; a.H = H; a.hd = hd; a.sgn = sin_sign;
    const long nvec = (long)B * T * H * (hd / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(stream_grid(nvec)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("vaa_model_rope");
}

extern "C" int vaa_model_swiglu_fwd(const uint16_t* gate, const uint16_t* up, uint16_t* y, long n, void* stream) {
    using namespace vaa;
    if (!gate || !up || !y || n <= 0 || (n % 8) != 0) { set_error("vaa_model_swiglu_fwd: bad arguments (n must be a positive multiple of 8)"); return VAA_E_INVALID; }
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, gate, up, y, n / 8);
    return check_launch("vaa_model_swiglu_fwd");
}

extern "C" int vaa_model_swiglu_bwd(const uint16_t* dy, const uint16_t* gate, const uint16_t* up, uint16_t* dgate, uint16_t* dup, long n, void* stream) {
    using namespace vaa;
    if (!dy || !gate || !up || !dgate || !dup || n <= 0 || (n % 8) != 0) { set_error("vaa_model_swiglu_bwd: bad arguments"); return VAA_E_INVALID; }
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(stream_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, dy, gate, up, dgate, dup, n / 8);
    return check_launch("vaa_model_swiglu_bwd");
}

extern "C" int vaa_model_scale_add(const uint16_t* x, const uint16_t* a, const uint16_t* ls, uint16_t* out, long rows, int D, void* stream) {
    using namespace vaa;
    if (!a || !ls || !out || rows <= 0 || D <= 0 || (D % 8) != 0) { set_error("vaa_model_scale_add: bad arguments (D must be a positive multiple of 8)"); return VAA_E_INVALID; }
    const long nvec = rows * (long)(D / 8);
    hipLaunchKernelGGL(scale_add_kernel, dim3(stream_grid(nvec)), dim3(256), 0, (hipStream_t)stream, x, a, ls, out, nvec, D / 8);
    return check_launch("vaa_model_scale_add");
}

extern "C" int vaa_model_rmsnorm_fwd(const uint16_t* x, const uint16_t* w, uint16_t* h, float* rstd, long rows, int D, float eps, void* stream) {
    using namespace vaa;
    if (!x || !w || !h || !rstd || rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * kNormThreads * kNormMaxVec) {
        set_error("vaa_model_rmsnorm_fwd: bad arguments (D must be a multiple of 8, <= 8192)");
        return VAA_E_INVALID;
    }
    hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, x, w, h, rstd, D, eps);
    return check_launch("vaa_model_rmsnorm_fwd");
}

extern "C" int vaa_model_rmsnorm_bwd(const uint16_t* gh, const uint16_t* gpass, const uint16_t* x, const uint16_t* w, const float* rstd, uint16_t* gx,
                                     long rows, int D, void* stream) {
    using namespace vaa;
    if (!gh || !x || !w || !rstd || !gx || rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * kNormThreads * kNormMaxVec) {
        set_error("vaa_model_rmsnorm_bwd: bad arguments");
        return VAA_E_INVALID;
    }
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, gh, gpass, x, w, rstd, gx, D);
    return check_launch("vaa_model_rmsnorm_bwd");
}

extern "C" int vaa_model_layernorm_fwd(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* h, float* stats, long rows, int D, float eps,
                                       void* stream) {
    using namespace vaa;
    if (!x || !w || !b || !h || !stats || rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * kNormThreads * kNormMaxVec) {
        set_error("vaa_model_layernorm_fwd: bad arguments (D must be a multiple of 8, <= 8192)");
        return VAA_E_INVALID;
    }
    const dim3 wgrid((unsigned)((rows + kNormWaveRows - 1) / kNormWaveRows)), wblk(64 * kNormWaveRows);
    if (!ln_wave_enabled()) hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, x, w, b, h, stats, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_fwd_wave_kernel<2>, wgrid, wblk, 0, (hipStream_t)stream, x, w, b, h, stats, rows, D, eps);
    else if (D <= 1536) hipLaunchKernelGGL(layernorm_fwd_wave_kernel<3>, wgrid, wblk, 0, (hipStream_t)stream, x, w, b, h, stats, rows, D, eps);
    else hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, x, w, b, h, stats, D, eps);
    return check_launch("vaa_model_layernorm_fwd");
}

extern "C" int vaa_model_layernorm_bwd(const uint16_t* gh, const uint16_t* gpass, const uint16_t* x, const uint16_t* w, const float* stats,
                                       uint16_t* gx, long rows, int D, void* stream) {
    using namespace vaa;
    if (!gh || !x || !w || !stats || !gx || rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * kNormThreads * kNormMaxVec) {
        set_error("vaa_model_layernorm_bwd: bad arguments");
        return VAA_E_INVALID;
    }
    const dim3 wgrid((unsigned)((rows + kNormWaveRows - 1) / kNormWaveRows)), wblk(64 * kNormWaveRows);
    if (!ln_wave_enabled()) hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, gh, gpass, x, w, stats, gx, D);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_bwd_wave_kernel<2>, wgrid, wblk, 0, (hipStream_t)stream, gh, gpass, x, w, stats, gx, rows, D);
    else if (D <= 1536) hipLaunchKernelGGL(layernorm_bwd_wave_kernel<3>, wgrid, wblk, 0, (hipStream_t)stream, gh, gpass, x, w, stats, gx, rows, D);
    else hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)rows), dim3(kNormThreads), 0, (hipStream_t)stream, gh, gpass, x, w, stats, gx, D);
    return check_launch("vaa_model_layernorm_bwd");
}
