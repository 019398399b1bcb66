// Micro-benchmark: cycles per ds_read_b128 for the A-fragment access pattern of maxsim_stream.hip under different
// row pitches.  hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_b128.hip -o /tmp/lds_b128 && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(int pitch, int kq_stride, int waves_active, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[150 * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 150 * 256; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = i;
    __syncthreads();
    const int fj = lane & 15, kq = lane >> 4;
    const char* base = MODE == 0 ? smem + fj * pitch + kq * kq_stride + wv * 1024   // row-per-lane pattern
                     : MODE == 1 ? smem + lane * 16 + wv * 1024                     // contiguous reference
                     : MODE == 2 ? smem + fj * 128 + ((kq ^ (fj >> 1)) & 7) * 16 + wv * 2048   // score_gemm swizzle (shipped)
                                 : smem + fj * 128 + ((kq ^ (((fj >> 1) & 3) << 1)) & 7) * 16 + wv * 2048;  // candidate
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wv < waves_active) {
        for (int it = 0; it < 200; ++it) {
            f32x4 a[16];
            const unsigned addr = (unsigned)(size_t)(base - smem) + (it & 1) * 16384;
#pragma unroll
            for (int mm = 0; mm < 16; ++mm)  // forced: one real ds_read_b128 each (the contents are loop-invariant)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[mm]) : "v"(addr), "n"(MODE >= 2 ? (mm & 7) * 2048 : mm * 64));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += a[0] + a[15];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4096);
    auto run = [&](const char* name, int mode, int pitch, int kqs, int waves) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, pitch, kqs, waves, d, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, pitch, kqs, waves, d, sink);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, pitch, kqs, waves, d, sink);
            else hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, pitch, kqs, waves, d, sink);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        }
        const double per = (double)h / (200.0 * 16 * waves);
        printf("%-46s waves=%d  %8llu ticks  %.2f ticks per wave-read (1 KiB)  -> %.0f B/tick/CU\n", name, waves, h, per,
               1024.0 / per);
    };
    for (int waves : {1, 4}) {
        run("contiguous (lane*16)", 1, 0, 0, waves);
        run("pitch 4112, kq*16 (shipped)", 0, 4112, 16, waves);
        run("pitch 4096 (no pad)", 0, 4096, 16, waves);
        run("pitch 4128 (+32)", 0, 4128, 16, waves);
        run("pitch 4160 (+64)", 0, 4160, 16, waves);
        run("pitch 4224 (+128)", 0, 4224, 16, waves);
        run("pitch 272 (16 rows x 256B + 16)", 0, 272, 16, waves);
        run("pitch 144 (128B + 16)", 0, 144, 16, waves);
        run("pitch 160 (128B + 32)", 0, 160, 16, waves);
        run("pitch 2080 (fp16 rows + 32)", 0, 2080, 16, waves);
        run("pitch 1056 (1 KiB + 32)", 0, 1056, 16, waves);
        run("128B rows, swizzle c^(r>>1) (score_gemm shipped)", 2, 0, 0, waves);
        run("128B rows, swizzle c^(((r>>1)&3)<<1)", 3, 0, 0, waves);
    }
    return 0;
}
