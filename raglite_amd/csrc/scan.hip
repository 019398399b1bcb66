// a6: streaming similarity scan, B <= 4 queries per corpus pass (VALU; the MFMA tile kernel in
// maxsim_stream.hip takes over for larger batches), plus the small helper kernels around it.
//
// Replaces the distance expression the reference hands to DuckDB / pgvector
// (src/raglite/_search.py:69-72, src/raglite/_typing.py:123-134): sim = 1 - dist with
//   cosine: dist = 1 - e.q/(|e||q|)      dot: dist = -(e.q)      l2: dist = |e - q|_2
// HBM-bound: algorithmic bytes = 4*N*dim per pass.  One wave per row: a
// dim = 1024 row is four coalesced 1-KiB `global_load_dwordx4`; the query lives in 16 VGPRs with the
// same column mapping; the per-row score is wave-uniform after a 6-step butterfly.
// The same kernel in SCAN_RAW_DOT mode is the query-adapter matvec out[b] = A @ q[b]
// (src/raglite/_search.py:62).
#include "common.h"

namespace rl {

// Streamed-once data (corpus rows): non-temporal loads skip the L1 allocation (measured on the LDS-DMA stream of the
// MaxSim kernel: issued -> landed 18 % shorter).
template <int VEC>
__device__ __forceinline__ void vload_nt(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        v[0] = __builtin_nontemporal_load(p);
    }
}

template <int VEC>
__device__ __forceinline__ void vload(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}

// Similarity from the reduced accumulator.  Every operation is a correctly-rounded IEEE fp32 op
// (hipcc default, no fast-math) so integer-valued test data gives bit-identical scores on the
// oracle's fp32 variant.
__device__ __forceinline__ float finish_score(float acc, float row_norm, float q_norm, int mode) {
    switch (mode) {
        case SCAN_COSINE: {
            const float c = acc / (row_norm * q_norm);
            return 1.0f - (1.0f - c);  // sim = 1 - dist, dist = 1 - cos (src/raglite/_search.py:72)
        }
        case SCAN_DOT: return 1.0f + acc;          // dist = -(e.q)
        case SCAN_L2: return 1.0f - sqrtf(acc);    // acc = sum (e-q)^2
        default: return acc;
    }
}

template <int NV, int VEC, int BQ>
__global__ __launch_bounds__(256) void scan_rows_kernel(const float* __restrict__ E, int64_t n, int dim,
                                                         const float* __restrict__ queries,
                                                         const float* __restrict__ row_norm, int mode,
                                                         float* __restrict__ scores, int64_t ld, const uint32_t* __restrict__ run_if) {
    if (run_if && *run_if == 0u) return;  // (a guarded launch: the full-precision pass behind a half-bytes search of a wide index)
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    int col[NV];
    bool ok[NV];
    float q[BQ][NV][VEC];
    float qn[BQ];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        col[v] = (v * 64 + lane) * VEC;
        ok[v] = col[v] < dim;
    }
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) q[b][v][j] = 0.f;
            if (ok[v]) vload<VEC>(queries + (int64_t)b * dim + col[v], q[b][v]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) ss = fmaf(q[b][v][j], q[b][v][j], ss);
        }
        qn[b] = sqrtf(wave_sum(ss));
    }
    const bool l2 = (mode == SCAN_L2);
    for (int64_t r = wave0; r < n; r += 2 * n_waves) {
        const int64_t r1 = r + n_waves;
        const bool has1 = r1 < n;
        float x0[NV][VEC], x1[NV][VEC];
        const float* p0 = E + r * (int64_t)dim;
        const float* p1 = E + (has1 ? r1 : r) * (int64_t)dim;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { x0[v][j] = 0.f; x1[v][j] = 0.f; }
            if (ok[v]) { vload_nt<VEC>(p0 + col[v], x0[v]); vload_nt<VEC>(p1 + col[v], x1[v]); }
        }
        float a0[BQ], a1[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (l2) {
                        const float t0 = x0[v][j] - q[b][v][j], t1 = x1[v][j] - q[b][v][j];
                        s0 = fmaf(t0, t0, s0);
                        s1 = fmaf(t1, t1, s1);
                    } else {
                        s0 = fmaf(x0[v][j], q[b][v][j], s0);
                        s1 = fmaf(x1[v][j], q[b][v][j], s1);
                    }
                }
            a0[b] = wave_sum(s0);
            a1[b] = wave_sum(s1);
        }
        const float rn0 = (mode == SCAN_COSINE) ? row_norm[r] : 1.f;
        const float rn1 = (mode == SCAN_COSINE && has1) ? row_norm[r1] : 1.f;
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            if (lane == b) {
                scores[(int64_t)b * ld + r] = finish_score(a0[b], rn0, qn[b], mode);
                if (has1) scores[(int64_t)b * ld + r1] = finish_score(a1[b], rn1, qn[b], mode);
            }
        }
    }
}

// ||e|| and ||e||^2 per row (fp32, one wave per row): precomputed once at index creation so that
// the cosine scan needs no second accumulator (4 B/row of extra traffic per pass).
template <int NV, int VEC>
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ E, int64_t n, int dim,
                                                         float* __restrict__ norm, float* __restrict__ sumsq) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += n_waves) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * VEC;
            if (c < dim) {
                float x[VEC];
                vload<VEC>(E + r * (int64_t)dim + c, x);
#pragma unroll
                for (int j = 0; j < VEC; ++j) ss = fmaf(x[j], x[j], ss);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            if (norm) norm[r] = sqrtf(ss);
            if (sumsq) sumsq[r] = ss;
        }
    }
}

__global__ __launch_bounds__(256) void cast_f16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                        int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const _Float16 h = (_Float16)src[i];
        uint16_t b;
        __builtin_memcpy(&b, &h, 2);
        dst[i] = b;
    }
}

__global__ __launch_bounds__(256) void fill_f32_kernel(float* __restrict__ dst, float value, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = value;
}

// In-place metric transform of raw dot products produced by the MFMA tile kernel (mode 1):
// scores[b*ld + i] = sim(e_i, q_b).  Query norms are recomputed per block (dim <= 4096 floats).
__global__ __launch_bounds__(256) void transform_kernel(float* __restrict__ scores, int64_t n, int64_t ld,
                                                         const float* __restrict__ row_norm,
                                                         const float* __restrict__ row_sumsq,
                                                         const float* __restrict__ queries, int dim, int mode, float pre_scale,
                                                         const uint32_t* __restrict__ run_if) {
    __shared__ float part[4];
    if (run_if && *run_if == 0u) return;
    const int b = blockIdx.y;
    float ss = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float v = queries[(int64_t)b * dim + c];
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float qss = (part[0] + part[1]) + (part[2] + part[3]);
    const float qn = sqrtf(qss);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float d = scores[(int64_t)b * ld + i];
        const float out = transform_score(d * pre_scale, mode, mode == SCAN_COSINE ? row_norm[i] : 1.f, mode == SCAN_L2 ? row_sumsq[i] : 0.f, qn, qss);
        scores[(int64_t)b * ld + i] = out;
    }
}

template <int NV, int VEC, int BQ>
static int scan_t(const float* E, int64_t n, int32_t dim, const float* q, const float* rn, int mode, float* sc,
                  int64_t ld, hipStream_t s, const uint32_t* run_if) {
    const int blocks = persistent_grid(scan_rows_kernel<NV, VEC, BQ>, 256, (n + 7) / 8);
    hipLaunchKernelGGL((scan_rows_kernel<NV, VEC, BQ>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, q, rn, mode, sc,
                       ld, run_if);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

template <int NV, int VEC>
static int scan_nb(const float* E, int64_t n, int32_t dim, const float* q, int32_t nb, const float* rn, int mode,
                   float* sc, int64_t ld, hipStream_t s, const uint32_t* run_if) {
    // Queries are consumed 4 / 2 / 1 per corpus pass (registers permitting).
    constexpr bool wide = (NV * VEC <= 16);
    int32_t b = 0;
    while (b < nb) {
        const float* qb = q + (int64_t)b * dim;
        float* sb = sc + (int64_t)b * ld;
        if constexpr (wide) {
            if (nb - b >= 4) { RL_TRY((scan_t<NV, VEC, 4>(E, n, dim, qb, rn, mode, sb, ld, s, run_if))); b += 4; continue; }
            if (nb - b >= 2) { RL_TRY((scan_t<NV, VEC, 2>(E, n, dim, qb, rn, mode, sb, ld, s, run_if))); b += 2; continue; }
        }
        RL_TRY((scan_t<NV, VEC, 1>(E, n, dim, qb, rn, mode, sb, ld, s, run_if)));
        b += 1;
    }
    return RL_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_scan_rows(const float* E, int64_t n, int32_t dim, const float* queries, int32_t nb, const float* row_norm,
                     int mode, float* scores, int64_t ld, hipStream_t s, const uint32_t* run_if) {
    if (n <= 0 || nb <= 0) return RL_OK;
    const bool vec4 = (dim % 4 == 0) && aligned16(E) && aligned16(queries);
#define RL_SCAN(NV, VEC) return scan_nb<NV, VEC>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s, run_if)
    if (vec4) {
        const int nv = (dim + 255) / 256;
        if (nv <= 1) RL_SCAN(1, 4);
        if (nv <= 2) RL_SCAN(2, 4);
        if (nv <= 3) RL_SCAN(3, 4);
        if (nv <= 4) RL_SCAN(4, 4);
        if (nv <= 6) RL_SCAN(6, 4);
        if (nv <= 8) RL_SCAN(8, 4);
        if (nv <= 16) RL_SCAN(16, 4);
    } else {
        const int nv = (dim + 63) / 64;
        if (nv <= 1) RL_SCAN(1, 1);
        if (nv <= 2) RL_SCAN(2, 1);
        if (nv <= 4) RL_SCAN(4, 1);
        if (nv <= 8) RL_SCAN(8, 1);
        if (nv <= 16) RL_SCAN(16, 1);
        if (nv <= 32) RL_SCAN(32, 1);
    }
#undef RL_SCAN
    return fail(RL_ERR_UNSUPPORTED, "similarity scan: dim must be <= 4096 (multiple of 4) or <= 2048 otherwise");
}

template <int NV, int VEC>
static int norms_t(const float* E, int64_t n, int32_t dim, float* norm, float* sumsq, hipStream_t s) {
    const int blocks = persistent_grid(row_norms_kernel<NV, VEC>, 256, (n + 3) / 4);
    hipLaunchKernelGGL((row_norms_kernel<NV, VEC>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, norm, sumsq);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_row_norms(const float* E, int64_t n, int32_t dim, float* norm, float* sumsq, hipStream_t s) {
    if (n <= 0) return RL_OK;
    const bool vec4 = (dim % 4 == 0) && aligned16(E);
    if (vec4) {
        const int nv = (dim + 255) / 256;
        if (nv <= 1) return norms_t<1, 4>(E, n, dim, norm, sumsq, s);
        if (nv <= 2) return norms_t<2, 4>(E, n, dim, norm, sumsq, s);
        if (nv <= 4) return norms_t<4, 4>(E, n, dim, norm, sumsq, s);
        if (nv <= 8) return norms_t<8, 4>(E, n, dim, norm, sumsq, s);
        if (nv <= 16) return norms_t<16, 4>(E, n, dim, norm, sumsq, s);
    } else {
        const int nv = (dim + 63) / 64;
        if (nv <= 4) return norms_t<4, 1>(E, n, dim, norm, sumsq, s);
        if (nv <= 16) return norms_t<16, 1>(E, n, dim, norm, sumsq, s);
        if (nv <= 32) return norms_t<32, 1>(E, n, dim, norm, sumsq, s);
    }
    return fail(RL_ERR_UNSUPPORTED, "row norms: dim too large");
}

int launch_cast_f16(const float* src, uint16_t* dst, int64_t count, hipStream_t s) {
    if (count <= 0) return RL_OK;
    const int blocks = (int)std::min<int64_t>((count + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(cast_f16_kernel, dim3(blocks), dim3(256), 0, s, src, dst, count);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// dst = fp16(src * scale), rounded to nearest even: the HI plane of an fp32 corpus (api.hip: refresh_hi_plane)
__global__ __launch_bounds__(256) void cast_f16_scaled_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t groups, float scale) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto pk = [&](float x, float y) -> uint32_t {
        uint32_t w;
        const h2 t = (h2){(_Float16)(x * scale), (_Float16)(y * scale)};
        __builtin_memcpy(&w, &t, 4);
        return w;
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += stride) {
        const f4 a = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + 2 * i);
        const f4 b = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src) + 2 * i + 1);
        const uint4 o = make_uint4(pk(a[0], a[1]), pk(a[2], a[3]), pk(b[0], b[1]), pk(b[2], b[3]));
        reinterpret_cast<uint4*>(dst)[i] = o;
    }
}

int launch_cast_f16_scaled(const float* src, uint16_t* dst, int64_t count, float scale, hipStream_t s) {
    if (count <= 0) return RL_OK;
    if (count % 8 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t groups = count / 8;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((groups + 255) / 256, 8192));
    hipLaunchKernelGGL(cast_f16_scaled_kernel, dim3(blocks), dim3(256), 0, s, src, dst, groups, scale);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_fill_f32(float* dst, float value, int64_t count, hipStream_t s) {
    if (count <= 0) return RL_OK;
    const int blocks = (int)std::min<int64_t>((count + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, s, dst, value, count);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_transform(float* scores, int32_t nb, int64_t n, int64_t ld, const float* row_norm, const float* row_sumsq,
                     const float* queries, int32_t dim, int mode, hipStream_t s, float pre_scale, const uint32_t* run_if) {
    if (n <= 0 || nb <= 0) return RL_OK;
    const int bx = (int)std::min<int64_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(transform_kernel, dim3(bx, nb), dim3(256), 0, s, scores, n, ld, row_norm, row_sumsq, queries,
                       (int)dim, mode, pre_scale, run_if);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
