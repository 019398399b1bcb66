// How fast can one CU pull L2-resident (or HBM) bytes, by instruction kind?  Answers what bounds maxsim_pp.hip's operand stream.
//   mode 0: global_load_dwordx4 -> VGPRs          mode 1: global_load_lds_dwordx4 -> LDS (1 KiB per wave-instruction)
//   mode 2: global_load_lds_dword -> LDS (256 B per wave-instruction)
// Every workgroup (512 threads, one per CU) sweeps the same `region` bytes `iters` times in 1-KiB pieces, `waves` of its 8 waves issuing,
// each keeping `depth` instructions in flight.  hipcc --offload-arch=gfx950 -O3 l2_dma_rate.hip -o /tmp/l2_dma_rate && /tmp/l2_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void sweep(const char* __restrict__ buf, size_t region, int iters, int waves, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv >= waves) return;
    const uint32_t lds = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) + wv * 8192;
    const size_t pieces = region / 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t p = (size_t)wv + (size_t)blockIdx.x * 37;  // workgroups start at different pieces, then walk the region in step
    for (int it = 0; it < iters; ++it) {
        for (size_t k = 0; k < pieces / waves; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char* src = buf + ((p + (size_t)d * waves) % pieces) * 1024;
                const uint64_t s64 = (uint64_t)src;
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)s64), hi = __builtin_amdgcn_readfirstlane((uint32_t)(s64 >> 32));
                const char* us = (const char*)(((uint64_t)hi << 32) | lo);
                if constexpr (MODE == 0) {
                    f32x4 v;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(16u * lane), "s"(us) : "memory");
                    asm volatile("s_waitcnt vmcnt(%1)\n\tv_add_f32 %0, %0, %2" : "+v"(acc[0]) : "n"(DEPTH - 1), "v"(v[0]));
                } else if constexpr (MODE == 1) {
                    const uint32_t l = __builtin_amdgcn_readfirstlane(lds + (d & 7) * 1024);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_waitcnt vmcnt(%3)" ::"s"(l), "v"(16u * lane), "s"(us), "n"(DEPTH - 1) : "memory", "m0");
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t l = __builtin_amdgcn_readfirstlane(lds + (d & 7) * 1024 + q * 256);
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:%3\n\ts_waitcnt vmcnt(%4)" ::"s"(l), "v"(4u * lane), "s"(us), "n"(q * 256), "n"(4 * DEPTH - 1) : "memory", "m0");
                    }
                }
            }
            p += (size_t)DEPTH * waves;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}

template <int MODE, int DEPTH>
static double run(const char* buf, size_t region, int iters, int waves, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((sweep<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, buf, region, 1, waves, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((sweep<MODE, DEPTH>), dim3(256), dim3(512), 0, 0, buf, region, iters, waves, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const size_t pieces = region / 1024;
    const double bytes = 256.0 * iters * (double)(pieces / waves / DEPTH * DEPTH) * waves * 1024.0;
    return bytes / (ms * 1e-3) / 1e12;  // TB/s over the chip
}

int main() {
    const size_t big = (size_t)2 << 30;
    char* buf; float* sink;
    hipMalloc(&buf, big); hipMalloc(&sink, 64);
    hipMemset(buf, 1, big);
    const char* names[3] = {"global_load_dwordx4 -> VGPR", "global_load_lds_dwordx4", "global_load_lds_dword x4"};
    for (int waves : {8, 4, 2, 1}) {
        printf("L2-resident 1 MiB region, %d issuing wave(s) per CU, 8 in flight per wave:\n", waves);
        printf("  %-30s %6.2f TB/s\n", names[0], run<0, 8>(buf, 1 << 20, 64, waves, sink));
        printf("  %-30s %6.2f TB/s\n", names[1], run<1, 8>(buf, 1 << 20, 64, waves, sink));
        printf("  %-30s %6.2f TB/s\n", names[2], run<2, 8>(buf, 1 << 20, 64, waves, sink));
    }
    printf("L2-resident 1 MiB, 8 waves, 2 in flight per wave: %6.2f (VGPR) %6.2f (LDS x4) TB/s\n", run<0, 2>(buf, 1 << 20, 64, 8, sink), run<1, 2>(buf, 1 << 20, 64, 8, sink));
    printf("HBM 2 GiB region, 8 waves, 8 in flight:            %6.2f (VGPR) %6.2f (LDS x4) TB/s\n", run<0, 8>(buf, big, 1, 8, sink), run<1, 8>(buf, big, 1, 8, sink));
    printf("HBM 2 GiB region, 2 waves, 8 in flight:            %6.2f (VGPR) %6.2f (LDS x4) TB/s\n", run<0, 8>(buf, big, 1, 2, sink), run<1, 8>(buf, big, 1, 2, sink));
    return 0;
}
