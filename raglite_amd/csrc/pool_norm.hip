// a1 + a2 + a3: late-chunking mean-pool over token-row spans, L2 normalise, fp16 cast.
//
// Replaces src/raglite/_embed.py:131-140 (and the whole-string variant :154,158-164).  HBM-bound:
// algorithmic bytes = 4*T*dim read + (4 and/or 2)*S*dim written.  One wave per span; a token row of
// dim = 1024 is 4 KiB = four coalesced 1-KiB `global_load_dwordx4` per wave.  Sums are kept in fp64
// (the reference pools float64 arrays) -- at ~10 B/clk/CU of HBM the fp64 adds are <5 % of VALU time
// -- so the fp16 result is bit-identical to the reference except where the mean sits within 1e-16
// of a rounding boundary.
#include "common.h"

namespace rl {

template <int VEC>
struct VecLoad;
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        // token rows are read exactly once: non-temporal (no L1 allocation)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
};
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = __builtin_nontemporal_load(p); }
};

__device__ __forceinline__ uint16_t f64_to_f16_bits(double x) {
    const _Float16 h = (_Float16)x;  // single correctly-rounded (RNE) conversion
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

template <int NV, int VEC>
__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ tokens, int dim,
                                                         const int64_t* __restrict__ span_begin,
                                                         const int64_t* __restrict__ span_end, int64_t n_spans,
                                                         int normalize, double eps, float* __restrict__ out_f32,
                                                         uint16_t* __restrict__ out_f16) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    int col[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        col[v] = (v * 64 + lane) * VEC;
        ok[v] = col[v] < dim;
    }
    for (int64_t s = wave0; s < n_spans; s += n_waves) {
        const int64_t b = span_begin[s], e = span_end[s];
        double acc[NV][VEC];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[v][j] = 0.0;
        int64_t r = b;
        for (; r + 4 <= e; r += 4) {  // four rows (4*NV loads) in flight per lane
            float x0[NV][VEC], x1[NV][VEC], x2[NV][VEC], x3[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) {
                    VecLoad<VEC>::load(p0 + col[v], x0[v]);
                    VecLoad<VEC>::load(p0 + dim + col[v], x1[v]);
                    VecLoad<VEC>::load(p0 + 2 * (int64_t)dim + col[v], x2[v]);
                    VecLoad<VEC>::load(p0 + 3 * (int64_t)dim + col[v], x3[v]);
                }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {  // same row order as the two-row loop: bit-identical sums
                        acc[v][j] += (double)x0[v][j]; acc[v][j] += (double)x1[v][j];
                        acc[v][j] += (double)x2[v][j]; acc[v][j] += (double)x3[v][j];
                    }
        }
        for (; r + 2 <= e; r += 2) {  // two rows (2*NV loads) in flight per lane
            float x0[NV][VEC], x1[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
            const float* p1 = p0 + dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) { VecLoad<VEC>::load(p0 + col[v], x0[v]); VecLoad<VEC>::load(p1 + col[v], x1[v]); }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { acc[v][j] += (double)x0[v][j]; acc[v][j] += (double)x1[v][j]; }
        }
        if (r < e) {
            float x0[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) VecLoad<VEC>::load(p0 + col[v], x0[v]);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[v][j] += (double)x0[v][j];
        }
        const double n = (double)(e - b);  // n == 0 -> 0/0 = NaN like np.mean of zero rows
        double ss = 0.0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[v][j] = acc[v][j] / n;
                if (ok[v]) ss += acc[v][j] * acc[v][j];
            }
        if (normalize) {
            double norm = sqrt(wave_sum(ss));
            if (eps > 0.0) norm = fmax(norm, eps);
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[v][j] = acc[v][j] / norm;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!ok[v]) continue;
            if (out_f32) {
                float* o = out_f32 + s * (int64_t)dim + col[v];
                if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(o) =
                        make_float4((float)acc[v][0], (float)acc[v][1], (float)acc[v][2], (float)acc[v][3]);
                } else {
                    o[0] = (float)acc[v][0];
                }
            }
            if (out_f16) {
                uint16_t* o = out_f16 + s * (int64_t)dim + col[v];
                if constexpr (VEC == 4) {
                    ushort4 h;
                    h.x = f64_to_f16_bits(acc[v][0]); h.y = f64_to_f16_bits(acc[v][1]);
                    h.z = f64_to_f16_bits(acc[v][2]); h.w = f64_to_f16_bits(acc[v][3]);
                    *reinterpret_cast<ushort4*>(o) = h;
                } else {
                    o[0] = f64_to_f16_bits(acc[v][0]);
                }
            }
        }
    }
}

template <int NV, int VEC>
static int launch_t(const float* tokens, int32_t dim, const int64_t* sb, const int64_t* se, int64_t n_spans,
                    int normalize, double eps, float* o32, uint16_t* o16, hipStream_t s) {
    // Spans are ragged (4..60 token rows at cfg 4), so a few generations of workgroups balance better than one
    // persistent generation: 8x measured 2.34 ms vs 2.44 ms on 3.2 M token rows.
    const int blocks = (int)std::min<int64_t>((int64_t)persistent_grid(pool_norm_kernel<NV, VEC>, 256, (n_spans + 3) / 4) * 8,
                                              (n_spans + 3) / 4);
    hipLaunchKernelGGL((pool_norm_kernel<NV, VEC>), dim3(blocks), dim3(256), 0, s, tokens, (int)dim, sb, se, n_spans,
                       normalize, eps, o32, o16);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_pool_norm(const float* tokens, int32_t dim, const int64_t* sb, const int64_t* se, int64_t n_spans,
                     int normalize, double eps, float* o32, uint16_t* o16, hipStream_t s) {
    if (n_spans <= 0) return RL_OK;
    const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(tokens) & 15) == 0) &&
                      (!o32 || (reinterpret_cast<uintptr_t>(o32) & 15) == 0) &&
                      (!o16 || (reinterpret_cast<uintptr_t>(o16) & 7) == 0);
#define RL_POOL_CASE(NV, VEC) return launch_t<NV, VEC>(tokens, dim, sb, se, n_spans, normalize, eps, o32, o16, s)
    if (vec4) {
        const int nv = (dim + 255) / 256;
        if (nv <= 1) RL_POOL_CASE(1, 4);
        if (nv <= 2) RL_POOL_CASE(2, 4);
        if (nv <= 3) RL_POOL_CASE(3, 4);
        if (nv <= 4) RL_POOL_CASE(4, 4);
        if (nv <= 6) RL_POOL_CASE(6, 4);
        if (nv <= 8) RL_POOL_CASE(8, 4);
        if (nv <= 16) RL_POOL_CASE(16, 4);
    } else {
        const int nv = (dim + 63) / 64;
        if (nv <= 1) RL_POOL_CASE(1, 1);
        if (nv <= 2) RL_POOL_CASE(2, 1);
        if (nv <= 4) RL_POOL_CASE(4, 1);
        if (nv <= 8) RL_POOL_CASE(8, 1);
        if (nv <= 16) RL_POOL_CASE(16, 1);
        if (nv <= 32) RL_POOL_CASE(32, 1);
    }
#undef RL_POOL_CASE
    return fail(RL_ERR_UNSUPPORTED, "rl_pool_norm: dim must be <= 4096 (multiple of 4) or <= 2048 otherwise");
}

}  // namespace rl
