// Hardware probe (gfx950): lane/element maps of ds_read_b64_tr_b16 and mfma_f32_16x16x32_bf16, printed as tables.
// hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_tr_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

__global__ void tr_probe(short* out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int idx;
    if (mode == 0) idx = l * 4;                                  // natural: lane l -> its own 8 bytes
    else if (mode == 1) idx = (4 * g + (i >> 2)) * 136 + (i & 3) * 4;  // rows of a padded (136-short stride) matrix
    else idx = ((l * 7) % 64) * 4;                                // scrambled lanes
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + idx));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}

__device__ inline short f2bf(float f) { return (short)(__float_as_uint(f) >> 16); }

// A[i][k] = (i+1) + 0.01*... use small exactly representable integers: A[i][k] = (i == probe_i && k == probe_k), B[k][j] = k*16 + j (exact in bf16 up to 256? 511 needs 9 bits -> use k + 32*j/... )
__global__ void mfma_probe(float* out) {
    // D = A*B with A = selector of one k per row: A[i][k] = (k == i*2) ; B[k][j] = k + j/64.0 -> D[i][j] = 2i + j/64 (all exact in bf16: k<32 5 bits + 6 bits frac = too many) -> B[k][j] = k (j even) or -k (j odd), plus second run with B = j
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    for (int run = 0; run < 2; ++run) {
        v8s a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = g * 8 + j;  // hypothesis: lane (r, g) element j <-> k = 8g + j
            a[j] = f2bf(k == ((r * 2 + 1) & 31) ? 1.0f : 0.0f);      // A[i=r][k]
            b[j] = f2bf(run == 0 ? (float)k : (float)(r * 3 - 7));   // B[k][j=r]
        }
        v4f c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int q = 0; q < 4; ++q) out[run * 256 + l * 4 + q] = c[q];
    }
}

int main() {
    short* d; hipMalloc(&d, 256 * 2);
    short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        tr_probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("tr mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("\n"); }
    }
    float* df; hipMalloc(&df, 512 * 4);
    float hf[512];
    mfma_probe<<<1, 64>>>(df);
    hipMemcpy(hf, df, 2048, hipMemcpyDeviceToHost);
    // expected with C layout col = lane&15, row = (lane>>4)*4 + reg: run0 D[i][j] = (2i+1)&31 ; run1 D[i][j] = 3j-7
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
        const int col = l & 15, row = (l >> 4) * 4 + q;
        if (hf[l * 4 + q] != (float)((2 * row + 1) & 31)) ++bad;
        if (hf[256 + l * 4 + q] != (float)(3 * col - 7)) ++bad;
    }
    printf("mfma 16x16x32 bf16: A[i=l&15][k=8(l>>4)+j], B[k][j=l&15], C col=l&15 row=4(l>>4)+reg : %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    if (bad) for (int l = 0; l < 64; l += 5) printf("  lane %d: %g %g %g %g | %g %g %g %g\n", l, hf[l*4], hf[l*4+1], hf[l*4+2], hf[l*4+3], hf[256+l*4], hf[256+l*4+1], hf[256+l*4+2], hf[256+l*4+3]);
    return 0;
}
