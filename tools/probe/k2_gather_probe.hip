// k2_gather_probe.hip — the north-star formulation of K2 ("one lane per patch texel, inverse-affine candidate window, register
// accumulation, no atomics"), built to MEASURE it against the product's scatter (vaa_patch_grad.hip); not part of libvaa_hip.so.
//
// A thread owns one patch texel (v, u) and walks the images of its slot (b = s, s + S, ...). Per image it maps the texel's canvas
// position back through the pixel-space affine, bounds the output pixels whose source point can lie within one texel of it by the row
// sums of the inverse matrix (4 x 4 candidates for the reference's rotation <= 30 deg / shear <= 0.2), re-evaluates each candidate's
// source point with the exact fp32 chain of the forward, and — where the texel is one of the candidate's four corners — adds
// fl(G * weight) in (row, column) order: the reference's own per-texel accumulation order, in registers. S partial tiles are then added
// in fixed order. Deterministic by construction.
// Not covered (the product covers it): patches on the frame edge — out-of-frame source points clamp onto edge texels, which takes an
// unbounded ray/quadrant enumeration per edge texel; tools/k2_gather_probe.py draws interior placements only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         tools/probe/k2_gather_probe.hip -o tools/scratch/libk2gather.so        (tools/k2_gather_probe.py does it)
#include "../../roboticattack_amd/csrc/vaa_common.h"

using namespace vaa;

struct GatherArgs {
    const uint16_t* g;    // [B,6,224,224] bf16
    const int32_t* xy;    // [B,2]
    const float* theta;   // [B,6]
    const uint8_t* keep;  // [B,3,224*224/8]
    float* partial;       // [S][3*ph*pw]
    int B, ph, pw, S;
    float istd6[6];
};

__global__ __launch_bounds__(256) void k2_gather_kernel(GatherArgs a) {
    __shared__ float bgrid[VAA_IMG];
    if (threadIdx.x < VAA_IMG) bgrid[threadIdx.x] = base_coord(threadIdx.x);
    __syncthreads();
    const int plane = a.ph * a.pw;
    const int t = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (t >= plane) return;
    const int v = t / a.pw, u = t - v * a.pw;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int b = s; b < a.B; b += a.S) {
        const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
        float th[6];
#pragma unroll
        for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
        const PixAffine pa = pix_affine(th);
        const float det = pa.a00 * pa.a11 - pa.a01 * pa.a10;
        const float i00 = pa.a11 / det, i01 = -pa.a01 / det, i10 = -pa.a10 / det, i11 = pa.a00 / det;
        const float X = (float)(px + u) - pa.c0, Y = (float)(py + v) - pa.c1;
        const float jx = i00 * X + i01 * Y, iy = i10 * X + i11 * Y;        // the output pixel whose source point IS the texel
        const float rj = fabsf(i00) + fabsf(i01) + 1e-3f, ri = fabsf(i10) + fabsf(i11) + 1e-3f;  // |source - texel| < 1 per axis
        const int jlo = max(0, (int)ceilf(jx - rj)), jhi = min(VAA_IMG - 1, (int)floorf(jx + rj));
        const int ilo = max(0, (int)ceilf(iy - ri)), ihi = min(VAA_IMG - 1, (int)floorf(iy + ri));
        const uint16_t* gimg = a.g + (size_t)b * 6 * VAA_NPIX;
        const uint8_t* kimg = a.keep + (size_t)b * 3 * (VAA_NPIX / 8);
        float ab[3] = {0.0f, 0.0f, 0.0f};
        for (int i = ilo; i <= ihi; ++i)
            for (int j = jlo; j <= jhi; ++j) {
                int x0, y0;
                float w, n;
                sample_pos_frac(bgrid[j], bgrid[i], th, x0, y0, w, n);
                const int dx = px + u - x0, dy = py + v - y0;
                if ((unsigned)dx > 1u || (unsigned)dy > 1u) continue;
                const float wt = (dy ? n : 1.0f - n) * (dx ? w : 1.0f - w);  // so*e, so*w, n*e, n*w: the products of grid_sample's backward
                const int pix = i * VAA_IMG + j;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (!((kimg[(size_t)c * (VAA_NPIX / 8) + (pix >> 3)] >> (pix & 7)) & 1)) continue;
                    const float G = bf16_bits_to_f32(gimg[(size_t)c * VAA_NPIX + pix]) * a.istd6[c] +
                                    bf16_bits_to_f32(gimg[(size_t)(c + 3) * VAA_NPIX + pix]) * a.istd6[c + 3];
                    ab[c] += G * wt;
                }
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += ab[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) a.partial[((size_t)s * 3 + c) * plane + t] = acc[c];
}

__global__ __launch_bounds__(256) void k2_gather_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gpatch, int n, int S) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    double t = 0.0;
    for (int s = 0; s < S; ++s) t += (double)partial[(size_t)s * n + e];
    gpatch[e] = (float)t;
}

extern "C" int k2_gather_probe(const uint16_t* g, const int32_t* xy, const float* theta, const uint8_t* keep, int B, int ph, int pw, int S,
                               const float* std6, float* partial, float* gpatch, void* stream) {
    GatherArgs a;
    a.g = g; a.xy = xy; a.theta = theta; a.keep = keep; a.partial = partial; a.B = B; a.ph = ph; a.pw = pw; a.S = S;
    for (int q = 0; q < 6; ++q) a.istd6[q] = (float)(1.0 / (double)std6[q]);
    hipLaunchKernelGGL(k2_gather_kernel, dim3((ph * pw + 255) / 256, S), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k2_gather_reduce_kernel, dim3((3 * ph * pw + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, gpatch, 3 * ph * pw, S);
    return (int)hipGetLastError();
}
