// a6 over an fp16-STORED corpus (the reference stores fp16 embeddings,
// src/raglite/_embed.py:140, so this storage is lossless for real RAGLite data and halves the bytes of every
// HBM-bound pass).  Same contract as scan.hip: B <= 4 queries per pass on the VALU, fp32 arithmetic -- an fp16 value
// converts to fp32 exactly, so scores equal those of an fp32 index holding the same values up to summation order
// (8 elements per lane here instead of 4), and exactly for integer-valued data.
#include "common.h"

namespace rl {
namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load8h(const uint16_t* p, float (&v)[8]) {
    const h16x8 t = __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(p));  // 16 B, streamed once
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}

__device__ __forceinline__ float finish16(float acc, float row_norm, float q_norm, int mode) {
    switch (mode) {  // identical to scan.hip:finish_score
        case SCAN_COSINE: { const float c = acc / (row_norm * q_norm); return 1.0f - (1.0f - c); }
        case SCAN_DOT: return 1.0f + acc;
        case SCAN_L2: return 1.0f - sqrtf(acc);
        default: return acc;
    }
}

template <int NV, int BQ>
__global__ __launch_bounds__(256) void scan_rows16_kernel(const uint16_t* __restrict__ E, int64_t n, int dim,
                                                           const float* __restrict__ queries,
                                                           const float* __restrict__ row_norm, int mode,
                                                           float* __restrict__ scores, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    int col[NV];
    bool ok[NV];
    float q[BQ][NV][8];
    float qn[BQ];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        col[v] = (v * 64 + lane) * 8;
        ok[v] = col[v] < dim;
    }
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q[b][v][j] = 0.f;
            if (ok[v]) load8f(queries + (int64_t)b * dim + col[v], q[b][v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(q[b][v][j], q[b][v][j], ss);
        }
        qn[b] = sqrtf(wave_sum(ss));
    }
    const bool l2 = (mode == SCAN_L2);
    for (int64_t r = wave0; r < n; r += 2 * n_waves) {
        const int64_t r1 = r + n_waves;
        const bool has1 = r1 < n;
        float x0[NV][8], x1[NV][8];
        const uint16_t* p0 = E + r * (int64_t)dim;
        const uint16_t* p1 = E + (has1 ? r1 : r) * (int64_t)dim;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { x0[v][j] = 0.f; x1[v][j] = 0.f; }
            if (ok[v]) { load8h(p0 + col[v], x0[v]); load8h(p1 + col[v], x1[v]); }
        }
        float a0[BQ], a1[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
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
                scores[(int64_t)b * ld + r] = finish16(a0[b], rn0, qn[b], mode);
                if (has1) scores[(int64_t)b * ld + r1] = finish16(a1[b], rn1, qn[b], mode);
            }
        }
    }
}

// 1024 < dim <= 4096 (round 6), raw dots only: the approximate pass of a few-queries search over the HI plane of a wide fp32 index.  The two rows
// in flight stay PACKED (16 B = 8 halves per register quad) until they are multiplied -- converted up front, as above, a 4096-wide row pair
// alone would take 128 registers.  Same order of summation as above: lane l sums its elements v = 0 .. NV - 1, j = 0 .. 7, then the butterfly.
template <int NV, int BQ, bool L2>
__global__ __launch_bounds__(256) void scan_rows16_wide_kernel(const uint16_t* __restrict__ E, int64_t n, int dim, const float* __restrict__ queries,
                                                                const float* __restrict__ row_norm, int mode, float* __restrict__ scores, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float q[BQ][NV][8];
    float qn[BQ];
#pragma unroll
    for (int b = 0; b < BQ; ++b) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q[b][v][j] = 0.f;
            if ((v * 64 + lane) * 8 < dim) load8f(queries + (int64_t)b * dim + (v * 64 + lane) * 8, q[b][v]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ss = fmaf(q[b][v][j], q[b][v][j], ss);
        }
        qn[b] = sqrtf(wave_sum(ss));
    }
    constexpr bool l2 = L2;  // (mode == SCAN_L2; an fp16-STORED wide index scans its rows in every metric: the statements of scan_rows16_kernel)
    const h16x8 zero = (h16x8){0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t r = wave0; r < n; r += 2 * n_waves) {
        const int64_t r1 = r + n_waves;
        const bool has1 = r1 < n;
        const uint16_t* p0 = E + r * (int64_t)dim;
        const uint16_t* p1 = E + (has1 ? r1 : r) * (int64_t)dim;
        h16x8 t0[NV], t1[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 8;
            t0[v] = c < dim ? __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(p0 + c)) : zero;
            t1[v] = c < dim ? __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(p1 + c)) : zero;
        }
        float a0[BQ], a1[BQ];
#pragma unroll
        for (int b = 0; b < BQ; ++b) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (l2) {
                        const float d0 = (float)t0[v][j] - q[b][v][j], d1 = (float)t1[v][j] - q[b][v][j];
                        s0 = fmaf(d0, d0, s0);
                        s1 = fmaf(d1, d1, s1);
                    } else {
                        s0 = fmaf((float)t0[v][j], q[b][v][j], s0);
                        s1 = fmaf((float)t1[v][j], q[b][v][j], s1);
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
                scores[(int64_t)b * ld + r] = finish16(a0[b], rn0, qn[b], mode);
                if (has1) scores[(int64_t)b * ld + r1] = finish16(a1[b], rn1, qn[b], mode);
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void row_norms16_kernel(const uint16_t* __restrict__ E, int64_t n, int dim,
                                                           float* __restrict__ norm, float* __restrict__ sumsq) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += n_waves) {
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 64 + lane) * 8;
            if (c < dim) {
                float x[8];
                load8h(E + r * (int64_t)dim + c, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) ss = fmaf(x[j], x[j], ss);
            }
        }
        ss = wave_sum(ss);
        if (lane == 0) {
            if (norm) norm[r] = sqrtf(ss);
            if (sumsq) sumsq[r] = ss;
        }
    }
}

template <int NV, int BQ>
int scan16_t(const uint16_t* E, int64_t n, int32_t dim, const float* q, const float* rn, int mode, float* sc,
             int64_t ld, hipStream_t s) {
    const int blocks = persistent_grid(scan_rows16_kernel<NV, BQ>, 256, (n + 7) / 8);
    hipLaunchKernelGGL((scan_rows16_kernel<NV, BQ>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, q, rn, mode, sc, ld);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

template <int NV>
int scan16_nb(const uint16_t* E, int64_t n, int32_t dim, const float* q, int32_t nb, const float* rn, int mode,
              float* sc, int64_t ld, hipStream_t s) {
    int32_t b = 0;
    while (b < nb) {
        const float* qb = q + (int64_t)b * dim;
        float* sb = sc + (int64_t)b * ld;
        if (nb - b >= 4) { RL_TRY((scan16_t<NV, 4>(E, n, dim, qb, rn, mode, sb, ld, s))); b += 4; continue; }
        if (nb - b >= 2) { RL_TRY((scan16_t<NV, 2>(E, n, dim, qb, rn, mode, sb, ld, s))); b += 2; continue; }
        RL_TRY((scan16_t<NV, 1>(E, n, dim, qb, rn, mode, sb, ld, s)));
        b += 1;
    }
    return RL_OK;
}
template <int NV, int BQ>
int scan16_wide_t(const uint16_t* E, int64_t n, int32_t dim, const float* q, const float* rn, int mode, float* sc, int64_t ld, hipStream_t s) {
    if (mode == SCAN_L2) {
        const int blocks = persistent_grid(scan_rows16_wide_kernel<NV, BQ, true>, 256, (n + 7) / 8);
        hipLaunchKernelGGL((scan_rows16_wide_kernel<NV, BQ, true>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, q, rn, mode, sc, ld);
    } else {
        const int blocks = persistent_grid(scan_rows16_wide_kernel<NV, BQ, false>, 256, (n + 7) / 8);
        hipLaunchKernelGGL((scan_rows16_wide_kernel<NV, BQ, false>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, q, rn, mode, sc, ld);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}
template <int NV>
int scan16_wide_nb(const uint16_t* E, int64_t n, int32_t dim, const float* q, int32_t nb, const float* rn, int mode, float* sc, int64_t ld, hipStream_t s) {
    constexpr int BQ_MAX = NV <= 4 ? 4 : 2;  // queries per corpus pass, registers permitting (a lane keeps NV x 8 elements of every query)
    int32_t b = 0;
    while (b < nb) {
        const float* qb = q + (int64_t)b * dim;
        float* sb = sc + (int64_t)b * ld;
        if constexpr (BQ_MAX >= 4) {
            if (nb - b >= 4) { RL_TRY((scan16_wide_t<NV, 4>(E, n, dim, qb, rn, mode, sb, ld, s))); b += 4; continue; }
        }
        if (nb - b >= 2) { RL_TRY((scan16_wide_t<NV, 2>(E, n, dim, qb, rn, mode, sb, ld, s))); b += 2; continue; }
        RL_TRY((scan16_wide_t<NV, 1>(E, n, dim, qb, rn, mode, sb, ld, s)));
        b += 1;
    }
    return RL_OK;
}
}  // namespace

// dim % 8 == 0, 16-B aligned rows and queries (rl_index_create_f16 guarantees it); dim <= 4096.  Beyond 1024 (round 6): the approximate pass of a
// few-queries search over the HI plane of a WIDE fp32 index (api.hip: search_rows_hi; raw dots, half the bytes of the fp32 scan, which takes such
// an index one query per pass) and the row scan of an fp16-STORED wide index (every metric).
int launch_scan_rows16(const uint16_t* E, int64_t n, int32_t dim, const float* queries, int32_t nb,
                       const float* row_norm, int mode, float* scores, int64_t ld, hipStream_t s) {
    if (n <= 0 || nb <= 0) return RL_OK;
    if (dim % 8 != 0 || dim > 4096 || (mode == SCAN_COSINE && !row_norm)) return RL_ERR_UNSUPPORTED;
    if (dim <= 512) return scan16_nb<1>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s);
    if (dim <= 1024) return scan16_nb<2>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s);
    if (dim <= 2048) return scan16_wide_nb<4>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s);
    if (dim <= 3072) return scan16_wide_nb<6>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s);
    return scan16_wide_nb<8>(E, n, dim, queries, nb, row_norm, mode, scores, ld, s);
}

int launch_row_norms16(const uint16_t* E, int64_t n, int32_t dim, float* norm, float* sumsq, hipStream_t s) {
    if (n <= 0) return RL_OK;
    if (dim % 8 != 0 || dim > 4096) return RL_ERR_UNSUPPORTED;
    if (dim > 1024) {
        const int blocks = persistent_grid(row_norms16_kernel<8>, 256, (n + 3) / 4);
        hipLaunchKernelGGL((row_norms16_kernel<8>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, norm, sumsq);
    } else if (dim <= 512) {
        const int blocks = persistent_grid(row_norms16_kernel<1>, 256, (n + 3) / 4);
        hipLaunchKernelGGL((row_norms16_kernel<1>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, norm, sumsq);
    } else {
        const int blocks = persistent_grid(row_norms16_kernel<2>, 256, (n + 3) / 4);
        hipLaunchKernelGGL((row_norms16_kernel<2>), dim3(blocks), dim3(256), 0, s, E, n, (int)dim, norm, sumsq);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
