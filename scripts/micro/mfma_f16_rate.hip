// Micro-benchmark: issue rate of the two fp16 MFMA shapes on gfx950, one or two waves per SIMD, 1 block per CU on all CUs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_f16_rate.hip -o mfma_f16_rate ; run: ./mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: v_mfma_f32_16x16x32_f16 (NACC independent accumulators of 4 regs); 1: v_mfma_f32_32x32x16_f16 (NACC of 16 regs)
template <int SHAPE, int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void rate(float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    h16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane ^ i)); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float tot = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) tot += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) tot += acc[i][0] + acc[i][15];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * THREADS + threadIdx.x] = tot;
    if (threadIdx.x == 0 && blockIdx.x == 3) cyc[0] = t1 - t0;
}

template <int SHAPE, int NACC, int THREADS>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate<SHAPE, NACC, THREADS><<<256, THREADS>>>(out, 10, cyc);
    hipEventRecord(e0);
    rate<SHAPE, NACC, THREADS><<<256, THREADS>>>(out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma_per_wave = (double)iters * 8 * NACC;
    const double flop = 256.0 * (THREADS / 64) * n_mfma_per_wave * (SHAPE == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16);
    const double waves_per_simd = THREADS / 256.0;
    printf("%-44s %8.3f ms  %7.1f TF  %6.2f ticks/MFMA/SIMD (s_memtime)  clock-if-16|32cyc %.2f GHz\n", name, ms, flop / ms / 1e9,
           (double)c / (n_mfma_per_wave * waves_per_simd), (n_mfma_per_wave * waves_per_simd * (SHAPE == 0 ? 16 : 32)) / (ms * 1e6));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<0, 4, 256>("16x16x32 f16, 1 wave/SIMD, 4 acc", out, cyc);
    run<0, 8, 256>("16x16x32 f16, 1 wave/SIMD, 8 acc", out, cyc);
    run<0, 4, 512>("16x16x32 f16, 2 waves/SIMD, 4 acc", out, cyc);
    run<0, 8, 512>("16x16x32 f16, 2 waves/SIMD, 8 acc", out, cyc);
    run<1, 2, 256>("32x32x16 f16, 1 wave/SIMD, 2 acc", out, cyc);
    run<1, 4, 256>("32x32x16 f16, 1 wave/SIMD, 4 acc", out, cyc);
    run<1, 2, 512>("32x32x16 f16, 2 waves/SIMD, 2 acc", out, cyc);
    run<1, 4, 512>("32x32x16 f16, 2 waves/SIMD, 4 acc", out, cyc);
    return 0;
}
