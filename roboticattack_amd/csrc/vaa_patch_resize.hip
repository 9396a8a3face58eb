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

constexpr int kTapMax = 8;
#ifndef VAA_RESIZE_WGS
#define VAA_RESIZE_WGS 8192
#endif
// images a workgroup walks in sequence: 1 while the batch is small (all images in parallel, the fixed-order reduce adds them), more once
// the grid alone fills the chip (fewer partials)
static int resize_group(int B, int ph, int pw) {
    const long wgs_per_image = 3l * ((ph * pw + 255) / 256);
    long g = (long)B * wgs_per_image / VAA_RESIZE_WGS;
    return (int)(g < 1 ? 1 : (g > 8 ? 8 : g));
}
constexpr int kMaxCand = 12;

struct TapTab {
    float w[2][VAA_IMG][kTapMax];  // [axis: 0 = y, 1 = x][output position][tap]
    short lo[2][VAA_IMG], n[2][VAA_IMG];
};


// Tap tables of one image in LDS: one output position per thread (h + w positions). The weights are tap_weight's own fp32 values, so
// whoever reads the table computes exactly what the direct evaluation computes.
__device__ __forceinline__ void build_tap_tables(TapTab& tab, const Axis& ay, const Axis& ax, bool vert, bool horiz, int h, int w) {
    for (int t = threadIdx.x; t < h + w; t += blockDim.x) {
        const int axis = t < h ? 0 : 1, o = t < h ? t : t - h;
        if ((axis == 0 && !vert) || (axis == 1 && !horiz)) continue;
        const Axis& aa = axis == 0 ? ay : ax;
        const Taps tp = make_taps(aa, o);
        tab.lo[axis][o] = (short)tp.lo;
        tab.n[axis][o] = (short)tp.n;
        for (int q = 0; q < tp.n; ++q) tab.w[axis][o][q] = tap_weight(aa, tp, q);
    }
}

// grid = (B, 3, slices): workgroup (b, c, z) computes every slices-th chunk of output plane c of image b
__global__ __launch_bounds__(256) void patch_resize_fwd_kernel(ResizeArgs a) {
    __shared__ TapTab tab;
    const int b = blockIdx.x, c = blockIdx.y;
    const int h = a.pdesc[4 * b], w = a.pdesc[4 * b + 1];
    const float* src = a.patch + (size_t)c * a.ph * a.pw;
    float* dst = a.out + a.pdesc[4 * b + 2] + (size_t)c * h * w;
    const Axis ay = make_axis(a.ph, h), ax = make_axis(a.pw, w);
    const bool horiz = (w != a.pw), vert = (h != a.ph);
    const bool tabled = ay.max_interp <= kTapMax && ax.max_interp <= kTapMax && h <= VAA_IMG && w <= VAA_IMG;  // workgroup-uniform
    if (tabled) {
        build_tap_tables(tab, ay, ax, vert, horiz, h, w);
        __syncthreads();
    }
    for (int e = blockIdx.z * 256 + threadIdx.x; e < h * w; e += gridDim.z * 256) {
        const int oy = e / w, ox = e - oy * w;
        float o = 0.0f;
        if (tabled) {  // the same products and FMA chain as below, weights from the tables
            const int r0 = vert ? tab.lo[0][oy] : oy, nr = vert ? tab.n[0][oy] : 1;
            const int x0 = horiz ? tab.lo[1][ox] : ox, nx = horiz ? tab.n[1][ox] : 1;
            for (int r = 0; r < nr; ++r) {
                const float* srow = src + (size_t)(r0 + r) * a.pw + x0;
                float hv;
                if (horiz) {
                    hv = nx > 0 ? srow[0] * tab.w[1][ox][0] : 0.0f;
                    for (int j = 1; j < nx; ++j) hv = __builtin_fmaf(srow[j], tab.w[1][ox][j], hv);
                } else {
                    hv = srow[0];
                }
                if (!vert) { o = hv; break; }
                o = (r == 0) ? hv * tab.w[0][oy][0] : __builtin_fmaf(hv, tab.w[0][oy][r], o);
            }
            dst[e] = o;
            continue;
        }
        Taps tx, ty;
        if (horiz) tx = make_taps(ax, ox);
        if (vert) ty = make_taps(ay, oy);
        const int r0 = vert ? ty.lo : oy, nr = vert ? ty.n : 1;
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

// Adjoint, as a deterministic gather: element (y, x) of the base-patch gradient collects, image after image, every output (oy, ox)
// whose taps cover it. Candidate outputs come from the inverse of centre = scale*(o+0.5) with a +-2 margin and are tested exactly.
// grid = (image groups, 3, ceil(ph*pw / 256)): a thread owns ONE base element and walks the images of its group in order, so a batch of
// up to kResizeGroup images needs no partial buffer at all; larger batches leave one partial per group to the fixed-order reduce.
// Per image the workgroup first builds the tap tables of both axes in LDS (one output position per thread: the make_taps arithmetic —
// fp64 steps, a division per tap — is done h + w times per workgroup instead of ~20 times per element; 22 -> 9 us at config 5's
// per-rank shape), for taps of up to kTapMax entries (down-scaling by up to 3x); wider filters take the direct evaluation.
__device__ __forceinline__ float gather_direct(const ResizeArgs& a, const float* g, const Axis& ay, const Axis& ax, bool vert, bool horiz, int h, int w,
                                               int y, int x, int oy_lo, int oy_hi, int ox_lo, int ox_hi) {
    float acc = 0.0f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        float wy = 1.0f;
        if (vert) {
            const Taps ty = make_taps(ay, oy);
            if (y < ty.lo || y >= ty.lo + ty.n) continue;
            wy = tap_weight(ay, ty, y - ty.lo);
        }
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            float wx = 1.0f;
            if (horiz) {
                const Taps tx = make_taps(ax, ox);
                if (x < tx.lo || x >= tx.lo + tx.n) continue;
                wx = tap_weight(ax, tx, x - tx.lo);
            }
            acc += wx * wy * g[(size_t)oy * w + ox];  // torch: grad_in += wx*wy*grad_out, (oh, ow) scan order
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void patch_resize_bwd_kernel(ResizeArgs a, int group) {
    __shared__ TapTab tab;
    const int c = blockIdx.y, tid = threadIdx.x;
    const int e = blockIdx.z * 256 + tid;
    const bool live = e < a.ph * a.pw;
    const int y = live ? e / a.pw : 0, x = live ? e - y * a.pw : 0;
    const int b_lo = blockIdx.x * group, b_hi = min(a.B, b_lo + group);
    float total = 0.0f;
    for (int b = b_lo; b < b_hi; ++b) {
        const int h = a.pdesc[4 * b], w = a.pdesc[4 * b + 1];
        const float* g = a.gpacked + a.pdesc[4 * b + 2] + (size_t)c * h * w;
        const Axis ay = make_axis(a.ph, h), ax = make_axis(a.pw, w);
        const bool horiz = (w != a.pw), vert = (h != a.ph);
        const bool tabled = ay.max_interp <= kTapMax && ax.max_interp <= kTapMax && h <= VAA_IMG && w <= VAA_IMG;  // workgroup-uniform
        if (tabled) {
            __syncthreads();  // the previous image's tables are no longer read
            build_tap_tables(tab, ay, ax, vert, horiz, h, w);
            __syncthreads();
        }
        if (!live) continue;
        int oy_lo = y, oy_hi = y, ox_lo = x, ox_hi = x;
        if (vert) {
            oy_lo = max(0, (int)floorf(((float)y - ay.support - 1.0f) / ay.scale - 0.5f) - 2);
            oy_hi = min(h - 1, (int)ceilf(((float)y + ay.support + 1.0f) / ay.scale - 0.5f) + 2);
        }
        if (horiz) {
            ox_lo = max(0, (int)floorf(((float)x - ax.support - 1.0f) / ax.scale - 0.5f) - 2);
            ox_hi = min(w - 1, (int)ceilf(((float)x + ax.support + 1.0f) / ax.scale - 0.5f) + 2);
        }
        const int ncx = ox_hi - ox_lo + 1;
        if (!tabled || ncx > kMaxCand) {
            total += gather_direct(a, g, ay, ax, vert, horiz, h, w, y, x, oy_lo, oy_hi, ox_lo, ox_hi);
            continue;
        }
        float wxs[kMaxCand];  // the x weights do not depend on oy
#pragma unroll
        for (int q = 0; q < kMaxCand; ++q) {
            float wx = 0.0f;
            if (q < ncx) {
                wx = 1.0f;
                if (horiz) {
                    const int ox = ox_lo + q, d = x - tab.lo[1][ox];
                    wx = (d >= 0 && d < tab.n[1][ox]) ? tab.w[1][ox][d] : 0.0f;
                }
            }
            wxs[q] = wx;
        }
        float acc = 0.0f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            float wy = 1.0f;
            if (vert) {
                const int d = y - tab.lo[0][oy];
                if (d < 0 || d >= tab.n[0][oy]) continue;
                wy = tab.w[0][oy][d];
            }
#pragma unroll
            for (int q = 0; q < kMaxCand; ++q)
                if (q < ncx && wxs[q] != 0.0f) acc += wxs[q] * wy * g[(size_t)oy * w + ox_lo + q];  // torch: grad_in += wx*wy*grad_out, (oh, ow) scan order
        }
        total += acc;
    }
    if (live) a.out[((size_t)blockIdx.x * 3 + c) * a.ph * a.pw + e] = total;
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
    VAA_LAUNCH(patch_resize_fwd_kernel, dim3(B, 3, slices), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("vaa_patch_resize_fwd");
}

extern "C" size_t vaa_patch_resize_ws_bytes(int B, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0) return 0;
    const int group = vaa::resize_group(B, ph, pw), groups = (B + group - 1) / group;
    return groups > 1 ? (size_t)groups * 3 * ph * pw * sizeof(float) : 0;  // one partial per image group; a single group writes gpatch itself
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
    const size_t need = vaa_patch_resize_ws_bytes(B, ph, pw);
    if (need > 0 && (!ws || ws_bytes < need)) {
        set_error("vaa_patch_resize_bwd: workspace %zu B < required %zu B", ws_bytes, need);
        return VAA_E_WORKSPACE;
    }
    const int group = resize_group(B, ph, pw), groups = (B + group - 1) / group;
    ResizeArgs a;
    a.patch = nullptr; a.gpacked = gpacked; a.pdesc = pdesc; a.out = groups > 1 ? (float*)ws : gpatch; a.B = B; a.ph = ph; a.pw = pw;
    VAA_LAUNCH(patch_resize_bwd_kernel, dim3(groups, 3, (ph * pw + 255) / 256), dim3(256), 0, st, a, group);
    rc = check_launch("vaa_patch_resize_bwd");
    if (rc != VAA_OK || groups == 1) return rc;
    return launch_partial_reduce((const float*)ws, gpatch, 3 * ph * pw, groups, st, "vaa_patch_resize_bwd(reduce)");
}
