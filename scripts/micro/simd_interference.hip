// Micro-benchmark (round 4): what does an instruction of wave B cost the MFMA stream of wave A on the SAME SIMD?
// maxsim_pp_kernel's pieces ADD instead of overlapping (MFMAs 0.50 ms, + LDS fragment reads 0.18, + DMAs 0.25, + epilogue VALU 0.10),
// and moving the epilogue between the partner's MFMAs (STAG, profiles/r04_d_*) bought nothing.  This measures the model directly:
// a workgroup of 8 waves (two per SIMD); waves 0-3 ("A") issue back-to-back independent v_mfma_f32_16x16x32_f16, waves 4-7 ("B") issue
// N instructions of one kind per A-iteration.  Reported: time of A alone, B alone, both -- if both ~ max the kinds overlap, if both ~ sum
// they share the SIMD's time.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 simd_interference.hip -o simd_interference ; run: ./simd_interference
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

enum Kind { K_NONE = 0, K_VALU = 1, K_LDS = 2, K_DMA = 3, K_SALU = 4, K_MFMA = 5, K_VMEM = 6 };

// ROLE_A: waves 0-3 run the MFMA stream; ROLE_B: waves 4-7 run `per_iter` instructions of KIND per iteration
template <int KIND, bool ROLE_A, bool ROLE_B>
__global__ __launch_bounds__(512, 2) void interference(float* out, const char* src, int iters, int per_iter) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < 16 * 1024; i += 512) reinterpret_cast<float*>(smem)[i] = (float)(i & 63) * 0.001f;
    __syncthreads();
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    if (wv < 4) {
        if constexpr (!ROLE_A) return;
        h16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < iters; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);  // 32 per iteration
        }
        f32x4 s = acc[0];
        for (int i = 1; i < 8; ++i) s += acc[i];
        out[(blockIdx.x * 512 + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
    } else {
        if constexpr (!ROLE_B) return;
        float v0 = lane, v1 = lane * 2.f, v2 = lane * 3.f, v3 = lane * 4.f;
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0;
        uint32_t sacc = 0;
        const uint32_t rd = lds_base + 16u * lane + (wv - 4) * 8192;
        const char* gsrc = src + (size_t)(blockIdx.x & 15) * 65536;  // L2-resident: 1 MiB shared by all workgroups
        h16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
        f32x4 macc[4] = {r0, r0, r0, r0};
        for (int t = 0; t < iters; ++t) {
            for (int j = 0; j < per_iter; j += 4) {
                if constexpr (KIND == K_VALU) {
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                } else if constexpr (KIND == K_LDS) {
                    asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(rd) : "memory");
                } else if constexpr (KIND == K_DMA) {
                    const uint32_t l = __builtin_amdgcn_readfirstlane(lds_base + 32768 + (wv - 4) * 4096);
                    const uint32_t off = 16u * lane;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                                 "s_add_u32 m0, %0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                                 "s_add_u32 m0, %0, 2048\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                                 "s_add_u32 m0, %0, 3072\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                                 "s_waitcnt vmcnt(8)"
                                 ::"s"(l), "v"(off), "s"(gsrc + (size_t)((t * 7 + j) & 15) * 4096) : "memory", "m0", "scc");
                } else if constexpr (KIND == K_VMEM) {
                    const uint32_t off = 16u * lane;
                    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072\n\ts_waitcnt vmcnt(4)"
                                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(off), "s"(gsrc + (size_t)((t * 7 + j) & 15) * 4096) : "memory");
                } else if constexpr (KIND == K_SALU) {
                    asm volatile("s_add_u32 %0, %0, 1\n\ts_xor_b32 %0, %0, 5\n\ts_add_u32 %0, %0, 3\n\ts_xor_b32 %0, %0, 9" : "+s"(sacc)::"scc");
                } else if constexpr (KIND == K_MFMA) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) macc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, macc[i], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        out[(blockIdx.x * 512 + threadIdx.x)] = v0 + v1 + v2 + v3 + r0[0] + r1[1] + r2[2] + r3[3] + (float)sacc + macc[0][0] + macc[1][1] + macc[2][2] + macc[3][3];
    }
}

template <int KIND, bool A, bool B>
float run(float* out, const char* src, int iters, int per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((interference<KIND, A, B>), dim3(256), dim3(512), 0, 0, out, src, iters, per_iter);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((interference<KIND, A, B>), dim3(256), dim3(512), 0, 0, out, src, iters, per_iter);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <int KIND>
void report(const char* name, float* out, const char* src, int iters, int per_iter) {
    const float a = run<KIND, true, false>(out, src, iters, 0), b = run<KIND, false, true>(out, src, iters, per_iter),
                ab = run<KIND, true, true>(out, src, iters, per_iter);
    // cycles per instruction of B per SIMD at the clock A's alone-time implies (32 MFMAs x 16 cycles per iteration)
    const double cyc_per_ms = (double)iters * 32 * 16 / a;
    printf("%-28s per_iter %3d: A alone %.3f ms, B alone %.3f ms (%.1f cycles per B instruction), both %.3f ms = %.2f x max, %.2f x sum; B costs A %.1f cycles per instruction\n",
           name, per_iter, a, b, b * cyc_per_ms / ((double)iters * per_iter), ab, ab / (a > b ? a : b), ab / (a + b),
           (ab - a) * cyc_per_ms / ((double)iters * per_iter));
}

int main() {
    float* out;
    char* src;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    const int iters = 20000;
    printf("two waves per SIMD; A = 32 x v_mfma_f32_16x16x32_f16 per iteration (512 matrix-pipe cycles); B = per_iter instructions per iteration\n");
    for (int per : {12, 32, 64}) report<K_VALU>("VALU (v_fma_f32)", out, src, iters, per);
    for (int per : {12, 24}) report<K_LDS>("LDS (ds_read_b128)", out, src, iters, per);
    for (int per : {4, 12}) report<K_DMA>("DMA (global_load_lds x4)", out, src, iters, per);
    for (int per : {4, 12}) report<K_VMEM>("VMEM (global_load_dwordx4)", out, src, iters, per);
    for (int per : {32, 128}) report<K_SALU>("SALU (s_add / s_xor)", out, src, iters, per);
    for (int per : {16, 32}) report<K_MFMA>("MFMA (the same stream)", out, src, iters, per);
    return 0;
}
