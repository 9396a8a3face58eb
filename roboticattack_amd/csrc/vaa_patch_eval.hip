// vaa_patch_eval.hip — eval-time paste of an optimised patch onto rollout frames (uint8 in, uint8 out), batched.
//
// Replaces RandomPatchTransform.simulation_random_patch (appply_random_transform.py:43-78), which the LIBERO evaluation
// calls once per simulator frame (experiments/robot/libero/run_libero_eval_args_geo_batch.py:207): patch quantised to
// uint8 by torchvision ToPILImage (mul(255).byte()), pasted on a -100 canvas, optional fixed rotation+shear warp
// (F.affine_grid + F.grid_sample, same bit-exact coordinate chain as K1), composite where canvas >= 0, truncate to uint8.
//
// HBM-bound byte copy with a sparse footprint: a thread owns 16 pixels (48 B in, 48 B out, 16-byte accesses); items outside
// the conservative per-row footprint bounds are a straight copy.
#include "vaa_common.h"

namespace vaa {

struct EvalArgs {
    const uint8_t* img;
    const float* patch;
    const int32_t* xy;
    const float* theta;
    const int32_t* geometry;  // per image
    uint8_t* out;
    int B, ph, pw;
};

constexpr int kEvPix = 16, kEvItemsPerRow = VAA_IMG / kEvPix, kEvItemsPerImg = VAA_IMG * kEvItemsPerRow, kEvThreads = 256, kEvMaxRows = 20;

__device__ __forceinline__ float quant_u8(float p) { return (float)(uint8_t)(p * 255.0f); }  // ToPILImage: mul(255).byte()

struct QuantPatch {  // canvas_at() reads texels through operator[]: quantise on the fly
    const float* p;
    __device__ __forceinline__ float operator[](int idx) const { return quant_u8(p[idx]); }
    __device__ __forceinline__ QuantPatch operator+(int off) const { return QuantPatch{p + off}; }
};

__global__ __launch_bounds__(kEvThreads) void patch_apply_eval_kernel(const EvalArgs a) {
    __shared__ float bgrid[VAA_IMG];
    __shared__ short rb_lo[kEvMaxRows], rb_hi[kEvMaxRows];
    const int tid = threadIdx.x;
    const long item0 = (long)blockIdx.x * kEvThreads, total = (long)a.B * kEvItemsPerImg;
    const long grow0 = item0 / kEvItemsPerRow;
    if (tid < VAA_IMG) bgrid[tid] = base_coord(tid);
    if (tid < kEvMaxRows) {
        const long gr = grow0 + tid;
        int jlo = 0, jhi = -1;
        if (gr < (long)a.B * VAA_IMG) {
            const int b = (int)(gr / VAA_IMG), i = (int)(gr % VAA_IMG);
            const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
            if (a.geometry[b]) {
                float th[6];
#pragma unroll
                for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
                const PixAffine pa = pix_affine(th);
                const float xlo = (px == 0) ? -1e30f : (float)(px - 1), xhi = (px + a.pw == VAA_IMG) ? 1e30f : (float)(px + a.pw);
                const float ylo = (py == 0) ? -1e30f : (float)(py - 1), yhi = (py + a.ph == VAA_IMG) ? 1e30f : (float)(py + a.ph);
                float jl = -1e30f, jh = 1e30f;
                solve_interval(pa.a00, pa.a01 * (float)i + pa.c0, xlo, xhi, jl, jh);
                solve_interval(pa.a10, pa.a11 * (float)i + pa.c1, ylo, yhi, jl, jh);
                if (jl <= jh) { jlo = (int)fmaxf(0.0f, floorf(jl) - 1.0f); jhi = (int)fminf((float)(VAA_IMG - 1), ceilf(jh) + 1.0f); }
            } else if (i >= py && i < py + a.ph) {
                jlo = px;
                jhi = px + a.pw - 1;
            }
        }
        if (jhi < jlo) { jlo = 0; jhi = -1; }  // an interval wholly beyond the frame (side open to infinity) is an empty row
        rb_lo[tid] = (short)jlo;
        rb_hi[tid] = (short)jhi;
    }
    __syncthreads();
    const long item = item0 + tid;
    if (item >= total) return;
    const long grow = item / kEvItemsPerRow;
    const int j0 = (int)(item - grow * kEvItemsPerRow) * kEvPix;
    const int b = (int)(grow / VAA_IMG), i = (int)(grow - (long)b * VAA_IMG);
    const size_t base = ((size_t)grow * VAA_IMG + j0) * 3;
    const uint4* src = reinterpret_cast<const uint4*>(a.img + base);
    uint4 w[3] = {src[0], src[1], src[2]};
    const int rr = (int)(grow - grow0);
    if (rb_hi[rr] >= j0 && rb_lo[rr] <= j0 + kEvPix - 1) {
        uint8_t* bytes = reinterpret_cast<uint8_t*>(w);
        const int px = a.xy[2 * b], py = a.xy[2 * b + 1], plane = a.ph * a.pw;
        const bool geo = a.geometry[b] != 0;
        float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        if (geo) {
#pragma unroll
            for (int z = 0; z < 6; ++z) th[z] = a.theta[6 * b + z];
        }
        const QuantPatch qp{a.patch};
        for (int p = 0; p < kEvPix; ++p) {
            Samp s;
            if (geo) s = sample_pos(bgrid[j0 + p], bgrid[i], th);
            else { s.x0 = j0 + p; s.y0 = i; s.nw = 1.f; s.ne = 0.f; s.sw = 0.f; s.se = 0.f; }
            const int u0 = s.x0 - px, v0 = s.y0 - py;
            if (u0 < -1 || u0 >= a.pw || v0 < -1 || v0 >= a.ph) continue;
            for (int c = 0; c < 3; ++c) {
                const float cv = geo ? sample_canvas(qp + c * plane, a.ph, a.pw, px, py, s)
                                     : canvas_at(qp + c * plane, a.ph, a.pw, px, py, s.x0, s.y0);
                if (!(cv < 0.0f)) bytes[p * 3 + c] = (uint8_t)cv;  // torch.where(canvas < 0, image, canvas) -> astype(uint8)
            }
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(a.out + base);
    dst[0] = w[0];
    dst[1] = w[1];
    dst[2] = w[2];
}

}  // namespace vaa

extern "C" int vaa_patch_apply_eval(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta,
                                    const int32_t* geometry, int B, int ph, int pw, uint8_t* out_u8, void* stream) {
    using namespace vaa;
    if (B == 0) return VAA_OK;
    if (!img_u8 || !patch || !xy || !theta || !geometry || !out_u8) {
        set_error("vaa_patch_apply_eval: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0) {
        set_error("vaa_patch_apply_eval: bad sizes (B=%d ph=%d pw=%d)", B, ph, pw);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("vaa_patch_apply_eval: patch %dx%d larger than the frame", ph, pw);
        return VAA_E_UNSUPPORTED;
    }
    EvalArgs a;
    a.img = img_u8; a.patch = patch; a.xy = xy; a.theta = theta; a.geometry = geometry; a.out = out_u8; a.B = B; a.ph = ph; a.pw = pw;
    const long total = (long)B * kEvItemsPerImg;
    VAA_LAUNCH(patch_apply_eval_kernel, dim3((unsigned)((total + kEvThreads - 1) / kEvThreads)), dim3(kEvThreads), 0, (hipStream_t)stream, a);
    return check_launch("vaa_patch_apply_eval");
}
