// Micro-benchmark: what does one 16-row tile of the MaxSim kernel cost on a CU, piece by piece?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_loop.hip -o mfma_loop ; run: ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 1040, STAGE = 16 * PITCH;

// VARIANT bits: 1 = ds_read A fragments each tile, 2 = workgroup barrier each tile, 4 = write partials to LDS,
//               8 = single accumulator chain, 16 = use 32x32x2 MFMA instead of 16x16x4
template <int V>
__global__ __launch_bounds__(256, 1) void tile_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 2 * STAGE + 8192];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float qreg[2][64];
    for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 64; ++i) qreg[h][i] = (float)((lane * 7 + i * 3 + h) & 15) * 0.01f;
    for (int i = threadIdx.x; i < (4 * 2 * STAGE) / 4; i += 256) ((float*)smem)[i] = (float)(i & 31) * 0.001f;
    __syncthreads();
    const char* a_base = smem + w * 2 * STAGE + (lane & 15) * PITCH + (lane >> 4) * 16;
    f32x4 a[16];
    for (int mm = 0; mm < 16; ++mm) a[mm] = *(const f32x4*)(a_base + mm * 64);
    f32x4 tot = {0, 0, 0, 0};
    for (int t = 0; t < iters; ++t) {
        if constexpr (V & 1) {
            const char* ap = a_base + (t & 1) * STAGE;
#pragma unroll
            for (int mm = 0; mm < 16; ++mm) a[mm] = *(const f32x4*)(ap + mm * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (V & 16) {
            typedef float f32x16 __attribute__((ext_vector_type(16)));
            f32x16 big = {0};
#pragma unroll
            for (int mm = 0; mm < 16; ++mm)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) big = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mm][tt], qreg[0][4 * mm + tt], big, 0, 0, 0);
            acc[0] = (f32x4){big[0], big[1], big[2], big[3]};
        } else {
#pragma unroll
            for (int mm = 0; mm < 16; ++mm)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int hh = (V & 8) ? 0 : h;
                        acc[hh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mm][tt], qreg[h][4 * mm + tt], acc[hh], 0, 0, 0);
                    }
        }
        if constexpr (V & 4) {
            char* red = smem + 4 * 2 * STAGE + 0;
            *(f32x4*)(red + ((w * 2 + 0) * 64 + lane) * 16 % 8192) = acc[0];
        }
        if constexpr (V & 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        tot += acc[0] + acc[1];
        if constexpr (!(V & 1)) asm volatile("" : "+v"(a[0]), "+v"(a[7]));
    }
    *(f32x4*)(out + (blockIdx.x * 256 + threadIdx.x) * 4) = tot;
}

template <int V>
void run(const char* name, float* d_out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(tile_loop<V>, dim3(256), dim3(256), 0, 0, d_out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(tile_loop<V>, dim3(256), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_tile = ms * 1e3 / iters;
    printf("%-44s %8.3f us/tile  = %7.0f cycles @2.4GHz   (%.1f TF/s)\n", name, us_tile, us_tile * 2400,
           2.0 * 16 * 32 * 1024 * 256 / (us_tile * 1e-6) / 1e12);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 16);
    const int it = 4000;
    run<0>("mfma 16x16x4 x128, 2 acc", d, it);
    run<8>("mfma 16x16x4 x128, 1 acc (dependent)", d, it);
    run<16>("mfma 32x32x2 x64 (half the flops)", d, it);
    run<1>("+ ds_read A frags", d, it);
    run<3>("+ ds_read + barrier", d, it);
    run<7>("+ ds_read + barrier + partial write", d, it);
    run<2>("mfma + barrier only", d, it);
    return 0;
}
