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
    int B, ph, pw, geometry, mask_mode;
    Norm6 nrm;
};

constexpr int kGroupsPerRow = VAA_IMG / 8;                // 28
constexpr int kGroupsPerImg = VAA_IMG * kGroupsPerRow;    // 6272

__device__ __forceinline__ uint32_t norm_pack(float v, const Norm6& n, int c) {
    float o0 = (v - n.mean[c]) / n.stdv[c];
    float o1 = (v - n.mean[c + 3]) / n.stdv[c + 3];
    return f32_to_bf16_bits(o0) | (f32_to_bf16_bits(o1) << 16);
}

__global__ __launch_bounds__(256) void patch_apply_fwd_kernel(FwdArgs a) {
    __shared__ uint32_t lut[3][256];
    __shared__ float bgrid[VAA_IMG];
    const int tid = threadIdx.x;
    for (int e = tid; e < 768; e += 256) {
        int c = e >> 8, v = e & 255;
        float im = (float)v / 255.0f;  // torchvision ToTensor (appply_random_transform.py:108)
        lut[c][v] = norm_pack(im, a.nrm, c);
    }
    if (tid < VAA_IMG) bgrid[tid] = base_coord(tid);
    __syncthreads();

    const long total = (long)a.B * kGroupsPerImg;
    const int plane = a.ph * a.pw;
    for (long gid = (long)blockIdx.x * 256 + tid; gid < total; gid += (long)gridDim.x * 256) {
        const int b = (int)(gid / kGroupsPerImg);
        const int rem = (int)(gid - (long)b * kGroupsPerImg);
        const int i = rem / kGroupsPerRow;
        const int j0 = (rem - i * kGroupsPerRow) * 8;

        const uint2* src = reinterpret_cast<const uint2*>(a.img + ((size_t)(b * VAA_IMG + i) * VAA_IMG + j0) * 3);
        uint2 w0 = src[0], w1 = src[1], w2 = src[2];
        const uint32_t d[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};

        uint32_t L[3][8];  // packed {bf16 plane c, bf16 plane c+3} per pixel
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int k = p * 3 + c;
                L[c][p] = lut[c][(d[k >> 2] >> (8 * (k & 3))) & 0xffu];
            }

        const int px = a.xy[2 * b], py = a.xy[2 * b + 1];
        uint32_t kb[3] = {0u, 0u, 0u};
        if (a.geometry) {
            float th[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) th[q] = a.theta[6 * b + q];
            const float by = bgrid[i];
            // conservative reject: clamped source coords of the two end pixels bound those of the 6 in between
            Samp sa = sample_pos(bgrid[j0], by, th), sb = sample_pos(bgrid[j0 + 7], by, th);
            const int xmin = min(sa.x0, sb.x0), xmax = max(sa.x0, sb.x0) + 1;
            const int ymin = min(sa.y0, sb.y0), ymax = max(sa.y0, sb.y0) + 1;
            const bool maybe = !(xmax < px - 1 || xmin > px + a.pw || ymax < py - 1 || ymin > py + a.ph);
            if (maybe) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    Samp s = (p == 0) ? sa : ((p == 7) ? sb : sample_pos(bgrid[j0 + p], by, th));
                    const int u0 = s.x0 - px, v0 = s.y0 - py;
                    if (u0 < -1 || u0 >= a.pw || v0 < -1 || v0 >= a.ph) continue;  // all four corners off the patch
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float cv = sample_canvas(a.patch + c * plane, a.ph, a.pw, px, py, s);
                        if (keep_rule(cv, a.mask_mode)) {
                            L[c][p] = norm_pack(cv, a.nrm, c);
                            kb[c] |= 1u << p;
                        }
                    }
                }
            }
        } else {
            const int v = i - py;
            if ((unsigned)v < (unsigned)a.ph && j0 + 7 >= px && j0 < px + a.pw) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int u = j0 + p - px;
                    if ((unsigned)u >= (unsigned)a.pw) continue;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float cv = a.patch[c * plane + v * a.pw + u];
                        if (keep_rule(cv, a.mask_mode)) {
                            L[c][p] = norm_pack(cv, a.nrm, c);
                            kb[c] |= 1u << p;
                        }
                    }
                }
            }
        }

        const size_t obase = ((size_t)b * 6 * VAA_IMG + i) * VAA_IMG + j0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint4 lo, hi;
            lo.x = (L[c][0] & 0xffffu) | (L[c][1] << 16);
            lo.y = (L[c][2] & 0xffffu) | (L[c][3] << 16);
            lo.z = (L[c][4] & 0xffffu) | (L[c][5] << 16);
            lo.w = (L[c][6] & 0xffffu) | (L[c][7] << 16);
            hi.x = (L[c][0] >> 16) | (L[c][1] & 0xffff0000u);
            hi.y = (L[c][2] >> 16) | (L[c][3] & 0xffff0000u);
            hi.z = (L[c][4] >> 16) | (L[c][5] & 0xffff0000u);
            hi.w = (L[c][6] >> 16) | (L[c][7] & 0xffff0000u);
            *reinterpret_cast<uint4*>(a.out + obase + (size_t)c * VAA_NPIX) = lo;
            *reinterpret_cast<uint4*>(a.out + obase + (size_t)(c + 3) * VAA_NPIX) = hi;
        }
        if (a.keep) {
            const size_t kbase = ((size_t)b * 3 * VAA_NPIX + (size_t)i * VAA_IMG + j0) >> 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) a.keep[kbase + (size_t)c * (VAA_NPIX / 8)] = (uint8_t)kb[c];
        }
    }
}

}  // namespace vaa

extern "C" int vaa_patch_apply_fwd(const uint8_t* img_u8, const float* patch, const int32_t* xy, const float* theta, int B,
                                   int ph, int pw, int geometry, int mask_mode, const float* mean6, const float* std6,
                                   uint16_t* out_bf16, uint8_t* keep_bits, void* stream) {
    using namespace vaa;
    if (B == 0) return VAA_OK;  // empty batch: nothing to read or write (pointers may be null)
    if (!img_u8 || !patch || !xy || !out_bf16 || !mean6 || !std6 || (geometry && !theta)) {
        set_error("vaa_patch_apply_fwd: null pointer argument");
        return VAA_E_INVALID;
    }
    if (B < 0 || ph <= 0 || pw <= 0 || (mask_mode != VAA_MASK_LT_M20 && mask_mode != VAA_MASK_NE_M100)) {
        set_error("vaa_patch_apply_fwd: bad sizes/mode (B=%d ph=%d pw=%d mask_mode=%d)", B, ph, pw, mask_mode);
        return VAA_E_INVALID;
    }
    if (ph > VAA_IMG || pw > VAA_IMG) {
        set_error("vaa_patch_apply_fwd: patch %dx%d larger than the %dx%d frame", ph, pw, VAA_IMG, VAA_IMG);
        return VAA_E_UNSUPPORTED;
    }
    FwdArgs a;
    a.img = img_u8; a.patch = patch; a.xy = xy; a.theta = theta; a.out = out_bf16; a.keep = keep_bits;
    a.B = B; a.ph = ph; a.pw = pw; a.geometry = geometry ? 1 : 0; a.mask_mode = mask_mode;
    for (int q = 0; q < 6; ++q) { a.nrm.mean[q] = mean6[q]; a.nrm.stdv[q] = std6[q]; }
    const long total = (long)B * kGroupsPerImg;
    long blocks = (total + 511) / 512;  // two 8-pixel groups per thread amortise the per-workgroup LUT build
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(patch_apply_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("vaa_patch_apply_fwd");
}
