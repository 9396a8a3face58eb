// vaa_patch_resize.hip — resize_patch=True (BASELINE config 5): the per-image rescaling of the BASE patch and its adjoint.
//
// Replaces `patch = transforms.Resize((height, width))(patch)` of RandomPatchTransform.apply_random_patch_batch
// (appply_random_transform.py:113-116, semantics of SURVEY.md Appendix A-D2: every image scales the base patch) and its
// autograd backward. torchvision's Resize on a tensor is torch's antialiased bilinear interpolation
// (F.interpolate(mode='bilinear', antialias=True, align_corners=False)); the arithmetic below follows torch's CPU kernel
// (ATen UpSampleKernel.cpp: _compute_indices_min_size_weights_aa, horizontal pass then vertical pass, each output
// src[0]*w[0] followed by fma(src[j], w[j], acc)) operation for operation, so the forward is BIT-EXACT against it
// (tests/test_oracle_golden.py pins the C restatement, tests/test_gpu_kernels.py this kernel).
//
// One launch resizes the base patch for the whole batch into a packed buffer (image b: [3,h_b,w_b] at pdesc[b].offset);
// one launch + the fixed-order reduce of vaa_patch_grad.hip produce the adjoint. Launch counts do not depend on B.
// The horizontal intermediate is recomputed per output element instead of being staged (a 139x139 plane is 25 taps per
// element): the values are the same fp32 numbers torch stores in its temporary, so the result is unchanged.
#include "vaa_common.h"

namespace vaa {

struct Axis {
    float scale, support, invscale;
    int max_interp, in_size;
};

__device__ __forceinline__ Axis make_axis(int in_size, int out_size) {
    Axis ax;
    ax.in_size = in_size;
    ax.scale = (float)in_size / (float)out_size;  // area_pixel_compute_scale (align_corners=False, no scale_factor)
    ax.support = (ax.scale >= 1.0f) ? ax.scale : 1.0f;
    ax.invscale = (ax.scale >= 1.0f) ? (float)(1.0 / (double)ax.scale) : 1.0f;
    ax.max_interp = (int)ceilf(ax.support) * 2 + 1;
    return ax;
}

struct Taps {
    int lo, n;
    float center, total;
};

__device__ __forceinline__ float raw_weight(const Axis& ax, const Taps& t, int j) {  // HelperInterpLinear::aa_filter
    float x = (float)(((double)((float)(j + t.lo) - t.center) + 0.5) * (double)ax.invscale);
    x = fabsf(x);
    return (x < 1.0f) ? (float)(1.0 - (double)x) : 0.0f;
}

__device__ __forceinline__ Taps make_taps(const Axis& ax, int i) {
    Taps t;
    t.center = (float)((double)ax.scale * ((double)i + 0.5));
    long lo = (long)((double)(t.center - ax.support) + 0.5);
    lo = lo < 0 ? 0 : lo;
    long hi = (long)((double)(t.center + ax.support) + 0.5);
    hi = hi > ax.in_size ? ax.in_size : hi;
    long n = hi - lo;
    n = n < 0 ? 0 : (n > ax.max_interp ? ax.max_interp : n);
    t.lo = (int)lo;
    t.n = (int)n;
    t.total = 0.0f;
    for (int j = 0; j < t.n; ++j) t.total += raw_weight(ax, t, j);
    return t;
}

__device__ __forceinline__ float tap_weight(const Axis& ax, const Taps& t, int j) {
    const float w = raw_weight(ax, t, j);
    return (t.total != 0.0f) ? w / t.total : w;
}

struct ResizeArgs {
    const float* patch;    // [3,ph,pw] base patch (fwd) — or nullptr (bwd)
    const float* gpacked;  // bwd: d L / d packed
    const int32_t* pdesc;  // [B,4] = {h, w, offset, 0}
    float* out;            // fwd: packed; bwd: per-image base-patch gradients [B][3*ph*pw]
    int B, ph, pw;
};

// grid = (B, 3, slices): workgroup (b, c, z) computes every slices-th chunk of output plane c of image b
__global__ __launch_bounds__(256) void patch_resize_fwd_kernel(ResizeArgs a) {
    const int b = blockIdx.x, c = blockIdx.y;
    const int h = a.pdesc[4 * b], w = a.pdesc[4 * b + 1];
    const float* src = a.patch + (size_t)c * a.ph * a.pw;
    float* dst = a.out + a.pdesc[4 * b + 2] + (size_t)c * h * w;
    const Axis ay = make_axis(a.ph, h), ax = make_axis(a.pw, w);
    const bool horiz = (w != a.pw), vert = (h != a.ph);
    for (int e = blockIdx.z * 256 + threadIdx.x; e < h * w; e += gridDim.z * 256) {
        const int oy = e / w, ox = e - oy * w;
        Taps tx, ty;
        if (horiz) tx = make_taps(ax, ox);
        if (vert) ty = make_taps(ay, oy);
        const int r0 = vert ? ty.lo : oy, nr = vert ? ty.n : 1;
        float o = 0.0f;
        for (int r = 0; r < nr; ++r) {
            const float* srow = src + (size_t)(r0 + r) * a.pw;
            float hv;
            if (horiz) {
                hv = tx.n > 0 ? srow[tx.lo] * tap_weight(ax, tx, 0) : 0.0f;
                for (int j = 1; j < tx.n; ++j) hv = __builtin_fmaf(srow[tx.lo + j], tap_weight(ax, tx, j), hv);
            } else {
                hv = srow[ox];
            }
            if (!vert) { o = hv; break; }
            o = (r == 0) ? hv * tap_weight(ay, ty, 0) : __builtin_fmaf(hv, tap_weight(ay, ty, r), o);
        }
        dst[e] = o;
    }
}

// Adjoint, as a deterministic gather: element (y, x) of the base-patch gradient of image b collects every output (oy, ox) whose
// taps cover it. Candidate outputs come from the inverse of centre = scale*(o+0.5) with a +-2 margin and are tested exactly.
__global__ __launch_bounds__(256) void patch_resize_bwd_kernel(ResizeArgs a) {
    const int b = blockIdx.x, c = blockIdx.y;
    const int h = a.pdesc[4 * b], w = a.pdesc[4 * b + 1];
    const float* g = a.gpacked + a.pdesc[4 * b + 2] + (size_t)c * h * w;
    float* dst = a.out + ((size_t)b * 3 + c) * a.ph * a.pw;
    const Axis ay = make_axis(a.ph, h), ax = make_axis(a.pw, w);
    const bool horiz = (w != a.pw), vert = (h != a.ph);
    for (int e = blockIdx.z * 256 + threadIdx.x; e < a.ph * a.pw; e += gridDim.z * 256) {
        const int y = e / a.pw, x = e - y * a.pw;
        int oy_lo = y, oy_hi = y, ox_lo = x, ox_hi = x;
        if (vert) {
            oy_lo = max(0, (int)floorf(((float)y - ay.support - 1.0f) / ay.scale - 0.5f) - 2);
            oy_hi = min(h - 1, (int)ceilf(((float)y + ay.support + 1.0f) / ay.scale - 0.5f) + 2);
        }
        if (horiz) {
            ox_lo = max(0, (int)floorf(((float)x - ax.support - 1.0f) / ax.scale - 0.5f) - 2);
            ox_hi = min(w - 1, (int)ceilf(((float)x + ax.support + 1.0f) / ax.scale - 0.5f) + 2);
        }
        // the x taps do not depend on oy: evaluate them once (up to kMaxCand candidates in registers; a wider window — extreme
        // down-scaling — takes the generic path that recomputes them per row)
        constexpr int kMaxCand = 12;
        float wxs[kMaxCand];
        const int ncx = ox_hi - ox_lo + 1;
        const bool cached = ncx <= kMaxCand;
        if (cached) {
#pragma unroll
            for (int q = 0; q < kMaxCand; ++q) {
                float wx = 0.0f;
                if (q < ncx) {
                    wx = 1.0f;
                    if (horiz) {
                        const Taps tx = make_taps(ax, ox_lo + q);
                        wx = (x >= tx.lo && x < tx.lo + tx.n) ? tap_weight(ax, tx, x - tx.lo) : 0.0f;
                    }
                }
                wxs[q] = wx;
            }
        }
        float acc = 0.0f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            float wy = 1.0f;
            if (vert) {
                const Taps ty = make_taps(ay, oy);
                if (y < ty.lo || y >= ty.lo + ty.n) continue;
                wy = tap_weight(ay, ty, y - ty.lo);
            }
            if (cached) {
#pragma unroll
                for (int q = 0; q < kMaxCand; ++q)
                    if (q < ncx && wxs[q] != 0.0f) acc += wxs[q] * wy * g[(size_t)oy * w + ox_lo + q];  // torch: grad_in += wx*wy*grad_out, (oh, ow) scan order
                continue;
            }
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float wx = 1.0f;
                if (horiz) {
                    const Taps tx = make_taps(ax, ox);
                    if (x < tx.lo || x >= tx.lo + tx.n) continue;
                    wx = tap_weight(ax, tx, x - tx.lo);
                }
                acc += wx * wy * g[(size_t)oy * w + ox];
            }
        }
        dst[e] = acc;
    }
}

static int check_resize_args(const char* who, const void* p0, const int32_t* pdesc, const void* out, int B, int ph, int pw) {
    if (!p0 || !pdesc || !out) {
        set_error("%s: null pointer argument", who);
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0) {
        set_error("%s: bad sizes (B=%d ph=%d pw=%d)", who, B, ph, pw);
        return VAA_E_INVALID;
    }
    return VAA_OK;
}

}  // namespace vaa

extern "C" int vaa_patch_resize_fwd(const float* patch, int ph, int pw, const int32_t* pdesc, int B, float* packed, void* stream) {
    using namespace vaa;
    if (B == 0) return VAA_OK;
    int rc = check_resize_args("vaa_patch_resize_fwd", patch, pdesc, packed, B, ph, pw);
    if (rc != VAA_OK) return rc;
    ResizeArgs a;
    a.patch = patch; a.gpacked = nullptr; a.pdesc = pdesc; a.out = packed; a.B = B; a.ph = ph; a.pw = pw;
    const int slices = B * 3 >= 256 ? 4 : 32;  // few images: more workgroups per plane so the launch still spreads over the chip
    hipLaunchKernelGGL(patch_resize_fwd_kernel, dim3(B, 3, slices), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("vaa_patch_resize_fwd");
}

extern "C" size_t vaa_patch_resize_ws_bytes(int B, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0) return 0;
    return (size_t)B * 3 * ph * pw * sizeof(float);
}

extern "C" int vaa_patch_resize_bwd(const float* gpacked, int ph, int pw, const int32_t* pdesc, int B, float* gpatch, void* ws,
                                    size_t ws_bytes, void* stream) {
    using namespace vaa;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0 && gpatch && ph > 0 && pw > 0) {
        if (hipMemsetAsync(gpatch, 0, (size_t)3 * ph * pw * sizeof(float), st) != hipSuccess) return check_launch("vaa_patch_resize_bwd(memset)");
        return VAA_OK;
    }
    int rc = check_resize_args("vaa_patch_resize_bwd", gpacked, pdesc, gpatch, B, ph, pw);
    if (rc != VAA_OK) return rc;
    if (!ws || ws_bytes < vaa_patch_resize_ws_bytes(B, ph, pw)) {
        set_error("vaa_patch_resize_bwd: workspace %zu B < required %zu B", ws_bytes, vaa_patch_resize_ws_bytes(B, ph, pw));
        return VAA_E_WORKSPACE;
    }
    ResizeArgs a;
    a.patch = nullptr; a.gpacked = gpacked; a.pdesc = pdesc; a.out = (float*)ws; a.B = B; a.ph = ph; a.pw = pw;
    const int slices = B * 3 >= 256 ? 4 : 32;
    hipLaunchKernelGGL(patch_resize_bwd_kernel, dim3(B, 3, slices), dim3(256), 0, st, a);
    rc = check_launch("vaa_patch_resize_bwd");
    if (rc != VAA_OK) return rc;
    return launch_partial_reduce((const float*)ws, gpatch, 3 * ph * pw, B, st, "vaa_patch_resize_bwd(reduce)");
}
