// vaa_patch_grad.hip — K2: dL/d patch from the model's bf16 pixel gradient, one launch (+ a small reduce) per batch.
//
// Replaces the autograd backward of RandomPatchTransform.apply_random_patch_batch reached from
// `MSE_Distance.backward()` (UADA.py:148): bf16->f32 cast, (g/std0 + g/std1), torch.where mask,
// grid_sample backward (bilinear scatter, padding 'border'), canvas slice, sum over the batch.
//
// Design for gfx950:
//   * only the warped footprint of the patch is touched: per image and output row the kernel bounds the
//     columns whose source point can land on the pasted rectangle (inverse of the pixel-space affine,
//     +-1 px margin), prefix-sums the 224 row lengths in LDS and walks the flattened footprint with all
//     256 lanes busy; the exact forward coordinates are then recomputed per pixel, so bounds only need to
//     be conservative.
//   * each output pixel scatters G*w into a patch-shaped accumulator tile in LDS (ds_add_f64). fp64
//     accumulation makes the sum independent of arrival order to ~1e-16, i.e. the fp32 result is
//     run-to-run reproducible without serialising the scatter (the reference's CUDA path is not).
//   * workgroups are persistent over images (b = blockIdx.x, += gridDim.x) and accumulate every image they
//     own into the same tile (the tile is in patch coordinates), then write ONE fp32 partial; a second tiny
//     kernel adds the <=512 partials in fixed order. No global atomics anywhere.
//   * border padding: out-of-frame source points clamp onto frame-edge canvas pixels; when the patch touches
//     the frame edge the row bounds are opened to infinity on that side, everything else is unchanged.
#include "vaa_common.h"

namespace vaa {

struct GradArgs {
    const uint16_t* g;
    const float* patch;
    const int32_t* xy;
    const float* theta;
    const uint8_t* keep;
    float* partial;
    int B, ph, pw, geometry, mask_mode;
    float std6[6];
};

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

template <typename ACC, int NCH>
__global__ __launch_bounds__(256) void patch_grad_scatter_kernel(GradArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    ACC* acc = reinterpret_cast<ACC*>(smem_raw);  // [NCH][ph][pw]
    __shared__ float bgrid[VAA_IMG];
    __shared__ int row_start[VAA_IMG + 1];
    __shared__ short row_jlo[VAA_IMG];

    const int tid = threadIdx.x;
    const int plane = a.ph * a.pw;
    const int c_base = (NCH == 1) ? blockIdx.y : 0;
    for (int e = tid; e < NCH * plane; e += 256) acc[e] = (ACC)0;
    if (tid < VAA_IMG) bgrid[tid] = base_coord(tid);
    __syncthreads();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
        float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        if (a.geometry) {
#pragma unroll
            for (int q = 0; q < 6; ++q) th[q] = a.theta[6 * b + q];
        }
        // ---- per-row column bounds of the footprint (conservative) ----
        if (tid < VAA_IMG) {
            const int i = tid;
            int jlo = 0, jhi = -1;
            if (a.geometry) {
                const PixAffine pa = pix_affine(th);
                // source x must fall in [px-1, px+pw) (corner x0 or x0+1 on the patch); a side that lies on the
                // frame edge also receives every clamped out-of-frame sample (padding_mode='border').
                const float xlo = (px == 0) ? -1e30f : (float)(px - 1);
                const float xhi = (px + a.pw == VAA_IMG) ? 1e30f : (float)(px + a.pw);
                const float ylo = (py == 0) ? -1e30f : (float)(py - 1);
                const float yhi = (py + a.ph == VAA_IMG) ? 1e30f : (float)(py + a.ph);
                float jl = -1e30f, jh = 1e30f;
                solve_interval(pa.a00, pa.a01 * (float)i + pa.c0, xlo, xhi, jl, jh);
                solve_interval(pa.a10, pa.a11 * (float)i + pa.c1, ylo, yhi, jl, jh);
                if (jl <= jh) {
                    jlo = (int)fmaxf(0.0f, floorf(jl) - 1.0f);
                    jhi = (int)fminf((float)(VAA_IMG - 1), ceilf(jh) + 1.0f);
                }
            } else if (i >= py && i < py + a.ph) {
                jlo = px;
                jhi = px + a.pw - 1;
            }
            row_jlo[i] = (short)jlo;
            row_start[i + 1] = max(0, jhi - jlo + 1);
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of 224 row lengths: 56 lanes x 4 rows
            int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
            if (tid < 56) { l0 = row_start[4 * tid + 1]; l1 = row_start[4 * tid + 2]; l2 = row_start[4 * tid + 3]; l3 = row_start[4 * tid + 4]; }
            int s = l0 + l1 + l2 + l3, incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int t = __shfl_up(incl, o, 64);
                if (tid >= o) incl += t;
            }
            const int excl = incl - s;
            if (tid < 56) {
                row_start[4 * tid + 1] = excl + l0;
                row_start[4 * tid + 2] = excl + l0 + l1;
                row_start[4 * tid + 3] = excl + l0 + l1 + l2;
                row_start[4 * tid + 4] = excl + s;
            }
            if (tid == 0) row_start[0] = 0;
        }
        __syncthreads();
        const int T = row_start[VAA_IMG];

        const uint16_t* gb = a.g + (size_t)b * 6 * VAA_NPIX;
        for (int t = tid; t < T; t += 256) {
            int lo = 0, hi = VAA_IMG;  // last row whose start <= t
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int mid = (lo + hi) >> 1;
                if (row_start[mid] <= t) lo = mid; else hi = mid;
            }
            const int i = lo;
            const int j = row_jlo[i] + (t - row_start[i]);
            Samp s;
            if (a.geometry) {
                s = sample_pos(bgrid[j], bgrid[i], th);
            } else {
                s.x0 = j; s.y0 = i; s.nw = 1.0f; s.ne = 0.0f; s.sw = 0.0f; s.se = 0.0f;
            }
            const int u0 = s.x0 - px, v0 = s.y0 - py;
            if (u0 < -1 || u0 >= a.pw || v0 < -1 || v0 >= a.ph) continue;
            const bool uin0 = u0 >= 0, uin1 = (u0 + 1 < a.pw) && (s.x0 + 1 < VAA_IMG);
            const bool vin0 = v0 >= 0, vin1 = (v0 + 1 < a.ph) && (s.y0 + 1 < VAA_IMG);
            const int pix = i * VAA_IMG + j;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                const int c = c_base + cc;
                bool kept;
                if (a.keep) {
                    kept = (a.keep[((size_t)(b * 3 + c) * VAA_NPIX + pix) >> 3] >> (pix & 7)) & 1;
                } else {
                    const float cv = a.geometry ? sample_canvas(a.patch + c * plane, a.ph, a.pw, px, py, s)
                                                : canvas_at(a.patch + c * plane, a.ph, a.pw, px, py, j, i);
                    kept = keep_rule(cv, a.mask_mode);
                }
                if (!kept) continue;
                const float G = bf16_bits_to_f32(gb[(size_t)c * VAA_NPIX + pix]) / a.std6[c] +
                                bf16_bits_to_f32(gb[(size_t)(c + 3) * VAA_NPIX + pix]) / a.std6[c + 3];
                ACC* t0 = acc + cc * plane + v0 * a.pw + u0;
                if (vin0 && uin0) atomicAdd(t0, (ACC)G * (ACC)s.nw);
                if (vin0 && uin1) atomicAdd(t0 + 1, (ACC)G * (ACC)s.ne);
                if (vin1 && uin0) atomicAdd(t0 + a.pw, (ACC)G * (ACC)s.sw);
                if (vin1 && uin1) atomicAdd(t0 + a.pw + 1, (ACC)G * (ACC)s.se);
            }
        }
        __syncthreads();  // row tables are rebuilt for the next image
    }
    float* dst = a.partial + ((size_t)blockIdx.x * 3 + c_base) * plane;
    for (int e = tid; e < NCH * plane; e += 256) dst[e] = (float)acc[e];
}

// gpatch[e] = sum_s partial[s][e] in fixed order (fp64 running sum), e over 3*ph*pw.
__global__ __launch_bounds__(256) void patch_grad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gpatch,
                                                                 int n, int nparts) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += (double)partial[(size_t)p * n + e];
    gpatch[e] = (float)s;
}

static int grad_grid(int B) { return B < 512 ? B : 512; }

}  // namespace vaa

extern "C" size_t vaa_patch_grad_ws_bytes(int B, int ph, int pw) {
    if (B <= 0 || ph <= 0 || pw <= 0) return 0;
    return (size_t)vaa::grad_grid(B) * 3 * (size_t)ph * pw * sizeof(float);
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
    if (!gout_bf16 || !xy || !std6 || !gpatch || (geometry && !theta) || (!keep_bits && !patch)) {
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
    const int n = 3 * ph * pw;
    hipStream_t st = (hipStream_t)stream;
    if (!ws || ws_bytes < vaa_patch_grad_ws_bytes(B, ph, pw)) {
        set_error("vaa_patch_grad_gather: workspace %zu B < required %zu B", ws_bytes, vaa_patch_grad_ws_bytes(B, ph, pw));
        return VAA_E_WORKSPACE;
    }
    GradArgs a;
    a.g = gout_bf16; a.patch = patch; a.xy = xy; a.theta = theta; a.keep = keep_bits; a.partial = (float*)ws;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) a.std6[q] = std6[q];
    const int G = grad_grid(B);
    const size_t plane = (size_t)ph * pw;
    const size_t lds_budget = 150 * 1024;  // 160 KiB per CU minus the static tables
    hipError_t e = hipSuccess;
    if (3 * plane * sizeof(double) <= 64 * 1024) {  // e.g. 50x50: 60,000 B, two workgroups per CU
        hipLaunchKernelGGL((patch_grad_scatter_kernel<double, 3>), dim3(G), dim3(256), 3 * plane * sizeof(double), st, a);
    } else if (plane * sizeof(double) <= lds_budget) {  // up to ~138x138: one channel per workgroup
        const size_t bytes = plane * sizeof(double);
        if (bytes > 64 * 1024)
            e = hipFuncSetAttribute((const void*)patch_grad_scatter_kernel<double, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) hipLaunchKernelGGL((patch_grad_scatter_kernel<double, 1>), dim3(G, 3), dim3(256), bytes, st, a);
    } else if (plane * sizeof(float) <= lds_budget) {  // up to ~195x195: fp32 accumulation
        const size_t bytes = plane * sizeof(float);
        if (bytes > 64 * 1024)
            e = hipFuncSetAttribute((const void*)patch_grad_scatter_kernel<float, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) hipLaunchKernelGGL((patch_grad_scatter_kernel<float, 1>), dim3(G, 3), dim3(256), bytes, st, a);
    } else {
        set_error("vaa_patch_grad_gather: patch %dx%d does not fit the LDS accumulator", ph, pw);
        return VAA_E_UNSUPPORTED;
    }
    if (e != hipSuccess) {
        set_error("vaa_patch_grad_gather: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return VAA_E_LAUNCH;
    }
    int rc = check_launch("vaa_patch_grad_gather(scatter)");
    if (rc != VAA_OK) return rc;
    hipLaunchKernelGGL(patch_grad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)ws, gpatch, n, G);
    return check_launch("vaa_patch_grad_gather(reduce)");
}
