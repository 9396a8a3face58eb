// cu_fetch_probe.hip — how many bytes per clock can ONE CU pull through its vector-memory path on gfx950, by destination (LDS-DMA vs
// registers), access pattern (8 rows x 128 B at an 8 KB stride per wave instruction — K3h's k-chunk of row-major operands — vs 1 KB contiguous),
// source (L2-resident 1 MB vs a 263 MB stream), waves issuing and with / without a workgroup barrier per chunk?
// Behind K3h's 128-row floor (profiles/r04_head_fused.txt): head_stats_kernel moves ~21 B/clk per CU whatever the lines hit.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_fetch_probe tools/probe/cu_fetch_probe.hip && /tmp/cu_fetch_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

constexpr int kWG = 251, kT = 512, kWaves = 8;
constexpr int kRowBytes = 8192;  // row stride of the row-major operands (D = 4096 bf16)

struct Args {
    const char* src;    // region the workgroup reads from
    size_t wg_stride;   // bytes between the regions of consecutive workgroups (0: everybody reads the same region)
    int instr_per_wave; // wave instructions (1 KB each) per wave in total
    int pattern;        // 0: 8 rows x 128 B at kRowBytes stride, k-chunk after k-chunk; 1: contiguous KBs
    int active_waves;   // waves that issue
    int barrier_every;  // s_barrier after this many instructions per wave (0: never)
    int stagger;        // per-workgroup k offset (K3h's)
    int asym;           // pattern 3, LDS-DMA: 24 / 8 instructions in flight for the weight / hidden-row waves instead of 12 / 12
    unsigned* sink;
};

// address of instruction j of wave w, lane l
__device__ __forceinline__ const char* addr_of(const Args& a, int w, int j, int lane) {
    const char* base = a.src + (size_t)blockIdx.x * a.wg_stride;
    const int g = j * a.active_waves + w;
    const int kstart = a.stagger ? (int)((blockIdx.x * 5u) & 63u) : 0;  // K3h's k order: workgroup w starts at chunk 5 w mod 64
    if (a.pattern == 1) return base + (((size_t)g + (size_t)kstart * 16) & 2047) * 1024 + lane * 16;
    if (a.pattern == 2) {  // K3h at 128 rows: per k-chunk 16 instructions of the workgroup's OWN weight slab, then 16 of the SHARED hidden rows
        const int chunk = ((g >> 5) + kstart) & 63, i = g & 31;
        const char* b2 = i < 16 ? base : a.src + (size_t)kWG * 128 * kRowBytes;  // hidden rows: one shared 1 MB region behind the slabs
        return b2 + (size_t)((i & 15) * 8 + (lane >> 3)) * kRowBytes + (size_t)chunk * 128 + (lane & 7) * 16;
    }
    if (a.pattern == 3) {  // the same bytes, but HALF of the waves fetch only the weight slab (HBM misses) and the other half only the hidden rows (L2 hits)
        const int hw = a.active_waves / 2, side = w >= hw, gg = j * hw + (w - side * hw);  // instruction index within the side's stream
        const int chunk = ((gg >> 4) + kstart) & 63, i = gg & 15;
        const char* b2 = side ? a.src + (size_t)kWG * 128 * kRowBytes : base;
        return b2 + (size_t)(i * 8 + (lane >> 3)) * kRowBytes + (size_t)chunk * 128 + (lane & 7) * 16;
    }
    // rows: instruction = rows 8 i .. 8 i + 7 of chunk c; 16 instructions (128 rows) per chunk
    const int chunk = ((g >> 4) + kstart), i = g & 15;
    return base + (size_t)(i * 8 + (lane >> 3)) * kRowBytes + (size_t)(chunk & 63) * 128 + (size_t)((chunk >> 6) & 1) * 128 * kRowBytes + (lane & 7) * 16;
}

template <bool DMA>
__global__ __launch_bounds__(kT) void probe(Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned acc = 0;
    if (w < a.active_waves) {
        constexpr int DEPTH = 12;  // instructions in flight per wave
        if constexpr (DMA) {
            auto stream = [&](auto depth_c, int slot0) {
                constexpr int DP = decltype(depth_c)::value;
                for (int j = 0; j < a.instr_per_wave; ++j) {
                    auto* dst = (__attribute__((address_space(3))) void*)(smem + (size_t)((slot0 + (j % DP)) * 1024));
                    __builtin_amdgcn_global_load_lds(addr_of(a, w, j, lane), dst, 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DP - 1) : "memory");
                    if (a.barrier_every && (j + 1) % a.barrier_every == 0) __builtin_amdgcn_s_barrier();
                }
            };
            if (a.pattern == 3 && a.asym) {  // weight waves keep 24 KB each in flight, hidden-row waves 8 KB: 4 x 24 + 4 x 8 = 128 KB of LDS
                if (w < a.active_waves / 2) stream(std::integral_constant<int, 24>{}, w * 24);
                else stream(std::integral_constant<int, 8>{}, 96 + (w - a.active_waves / 2) * 8);
            } else {
                stream(std::integral_constant<int, DEPTH>{}, w * DEPTH);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc = smem[tid * 16];
        } else {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 r[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = (u32x4){0, 0, 0, 0};
            for (int j0 = 0; j0 < a.instr_per_wave; j0 += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    acc += r[d][0];  // consume the value loaded DEPTH instructions ago: counted waits by the compiler
                    if (j0 + d < a.instr_per_wave) r[d] = *reinterpret_cast<const u32x4*>(addr_of(a, w, j0 + d, lane));
                    if (a.barrier_every && (j0 + d + 1) % a.barrier_every == 0) __builtin_amdgcn_s_barrier();
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += r[d][0];
        }
    } else if (a.barrier_every) {
        for (int j = 0; j < a.instr_per_wave; ++j)
            if ((j + 1) % a.barrier_every == 0) __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
#endif
}

static float run(bool dma, const Args& a) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds = 128 * 1024;
    if (dma) CHECK(hipFuncSetAttribute((const void*)probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else CHECK(hipFuncSetAttribute((const void*)probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        if (dma) hipLaunchKernelGGL(probe<true>, dim3(kWG), dim3(kT), lds, 0, a);
        else hipLaunchKernelGGL(probe<false>, dim3(kWG), dim3(kT), lds, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const size_t big = (size_t)kWG * 128 * kRowBytes;  // 263 MB: 128 rows of 8 KB per workgroup = K3h's weight slab
    char* buf;
    unsigned* sink;
    CHECK(hipMalloc(&buf, big + (2 << 20)));
    CHECK(hipMemset(buf, 1, big + (2 << 20)));
    CHECK(hipMalloc(&sink, 64));
    const int total_instr = 2048;  // per workgroup: 2 MB = K3h's 128-row case (1 MB of weights + 1 MB of hidden rows)
    printf("per workgroup %d KB, %d workgroups; B/clk at 2.4 GHz\n", total_instr, kWG);
    printf("%-9s %-10s %-8s %-6s %-8s %-4s %9s %9s\n", "dest", "pattern", "source", "waves", "barrier", "k", "us", "B/clk/CU");
    for (int dma = 1; dma >= 0; --dma)
        for (int pattern = 0; pattern < 4; ++pattern)
            for (int src = 0; src < 2; ++src)        // 0: the same 2 MB for everybody (L2 hits), 1: own 1 MB slab (HBM stream); pattern 2 = both
                for (int stag = 0; stag < 2; ++stag)
                    for (int waves = 8; waves >= 2; waves >>= 1)
                        for (int bar = 0; bar < 4; ++bar) {
                            if ((waves != 8 && (bar || stag == 0)) || (pattern >= 2 && src == 0)) continue;
                            Args a;
                            a.src = buf;
                            a.wg_stride = (src == 0 && pattern < 2) ? 0 : (size_t)128 * kRowBytes;
                            a.instr_per_wave = ((src == 1 && pattern < 2) ? total_instr / 2 : total_instr) / waves;
                            a.pattern = pattern;
                            a.active_waves = waves;
                            a.barrier_every = bar ? (2 << bar) : 0;  // one s_barrier per 4 / 8 / 16 instructions per wave (K3h: 4 = one k-chunk)
                            a.stagger = stag;
                            a.asym = 0;
                            a.sink = sink;
                            const size_t bytes = (size_t)a.instr_per_wave * waves * 1024;
                            const float us = run(dma, a);
                            printf("%-9s %-10s %-8s %-6d %-8s %-4s %9.1f %9.1f\n", dma ? "LDS-DMA" : "registers", pattern == 3 ? "K3h split" : (pattern == 2 ? "K3h mix" : (pattern ? "contig" : "rows8x128")),
                                   pattern >= 2 ? "HBM+L2" : (src == 0 ? "L2" : "HBM"), waves, bar == 0 ? "-" : (bar == 1 ? "per 4" : (bar == 2 ? "per 8" : "per 16")), stag ? "stag" : "-", us, (double)bytes / (us * 2400.0));
                        }
    for (int bar = 0; bar < 2; ++bar) {
        Args a;
        a.src = buf; a.wg_stride = (size_t)128 * kRowBytes; a.instr_per_wave = total_instr / 8; a.pattern = 3; a.active_waves = 8; a.barrier_every = bar ? 4 : 0;
        a.stagger = 1; a.asym = 1; a.sink = sink;
        const float us = run(true, a);
        printf("%-9s %-10s %-8s %-6d %-8s %-4s %9.1f %9.1f   (24 / 8 in flight)\n", "LDS-DMA", "K3h split", "HBM+L2", 8, bar ? "per 4" : "-", "stag", us, (double)total_instr * 1024 / (us * 2400.0));
    }
    return 0;
}
