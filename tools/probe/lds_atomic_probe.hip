// lds_atomic_probe.hip — what does one LDS atomic wave-instruction cost on gfx950, by type and address pattern?
// Decides K2's accumulator: fp64 LDS atomics (order-independent to 1e-16) vs integer fixed point vs a gather without atomics.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic_probe tools/probe/lds_atomic_probe.hip && /tmp/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kTile = 7500;  // 3 x 50 x 50
constexpr int kIters = 256;

template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
    atomicAdd(p, v);
}
template <>
__device__ __forceinline__ void lds_add<unsigned long long>(unsigned long long* p, unsigned long long v) {
    atomicAdd(p, v);
}

// PATTERN 0: lane-consecutive addresses (conflict-free), 1: rows of a rotated footprint (half-wave = 32 consecutive pixels of a
// row, ~0.9 texel per pixel, second half-wave one row down), 2: pseudo-random, 3: every lane the same address
template <typename T, int PATTERN>
__global__ __launch_bounds__(1024) void probe(T* out, int iters) {
    __shared__ T tile[kTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < kTile; e += 1024) tile[e] = (T)0;
    __syncthreads();
    uint32_t s = tid * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        int a;
        if (PATTERN == 0) {
            a = (it * 64 + wave * 448 + lane) % kTile;
        } else if (PATTERN == 1) {
            const int row = (it + wave * 3 + (lane >> 5)) % 49, col = (int)((lane & 31) * 0.9f) + (it & 7);
            a = row * 50 + col;
        } else if (PATTERN == 2) {
            s = s * 1664525u + 1013904223u;
            a = (s >> 8) % kTile;
        } else {
            a = (it * 7 + wave) % kTile;
        }
        lds_add<T>(&tile[a], (T)1);
    }
    __syncthreads();
    if (tid < 64) out[blockIdx.x * 64 + tid] = tile[tid * 100];
}

template <typename T, int PATTERN>
static void run(const char* name, int blocks) {
    T* out;
    hipMalloc(&out, sizeof(T) * 64 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<T, PATTERN><<<blocks, 1024>>>(out, kIters);
    hipDeviceSynchronize();
    // subtract the empty-loop launch: time at iters and 4*iters
    float ms1 = 0, ms4 = 0;
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) probe<T, PATTERN><<<blocks, 1024>>>(out, kIters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms1, e0, e1);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) probe<T, PATTERN><<<blocks, 1024>>>(out, 4 * kIters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms4, e0, e1);
    const double us_per_launch_delta = (ms4 - ms1) * 1e3 / 10.0;               // 3*kIters extra iterations
    const double wave_instr_per_cu = 3.0 * kIters * 16.0 * (blocks / 256.0);   // 16 waves per block
    const double cyc = us_per_launch_delta * 2400.0 / wave_instr_per_cu;       // at 2.4 GHz nominal
    printf("%-28s blocks/CU=%d  %.2f us per %d-iter delta  -> %.1f cycles per wave-instruction per CU\n", name, blocks / 256,
           us_per_launch_delta, 3 * kIters, cyc);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int blocks = 256 * bpc;
        run<double, 0>("f64 consecutive", blocks);
        run<double, 1>("f64 footprint-rows", blocks);
        run<double, 2>("f64 random", blocks);
        run<double, 3>("f64 same-address", blocks);
        run<float, 0>("f32 consecutive", blocks);
        run<float, 1>("f32 footprint-rows", blocks);
        run<float, 2>("f32 random", blocks);
        run<float, 3>("f32 same-address", blocks);
        run<unsigned int, 0>("u32 consecutive", blocks);
        run<unsigned int, 1>("u32 footprint-rows", blocks);
        run<unsigned int, 2>("u32 random", blocks);
        run<unsigned int, 3>("u32 same-address", blocks);
        run<unsigned long long, 0>("u64 consecutive", blocks);
        run<unsigned long long, 1>("u64 footprint-rows", blocks);
        run<unsigned long long, 2>("u64 random", blocks);
        run<unsigned long long, 3>("u64 same-address", blocks);
    }
    return 0;
}
