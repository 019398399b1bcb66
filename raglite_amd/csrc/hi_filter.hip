// Small kernels of the half-bytes single-query search (api.hip: search_rows_hi): a corpus pass over the HI halves only
// (2 B per element) ranks approximately, a rigorous error bound turns the approximate top of the list into a candidate set
// that provably contains the exact top-k, and the candidates are re-scored with the exact kernels.
//
// a6 + a7 of SURVEY.md section 8a (src/raglite/_search.py:69-79) for B <= 16 queries: results identical to the full-precision
// pass, bit for bit.
#include "common.h"

namespace rl {
namespace {

// One block per query.  topk[b * k + j] = the k best APPROXIMATE similarities, descending.  With |approx - exact| <= m for every
// row, a row can be in the exact top-k only if its approximate score is >= (k-th best approximate) - 2 m =: thr[b]:
//   cosine: m = m_rel (the scores are cosines);  dot: m = m_rel * e_norm_bound * |q|.
// An unusable k-th score (NaN: fewer than k comparable rows) sets *flag: the caller's guarded full-precision pass then runs.
__global__ __launch_bounds__(256) void approx_threshold_kernel(const float* __restrict__ topk, int32_t nb, int32_t k,
                                                                const float* __restrict__ queries, int dim, int mode, float m_rel,
                                                                float e_norm_bound, float* __restrict__ thr, uint32_t* __restrict__ cnt,
                                                                uint32_t* __restrict__ flag) {
    // ONE block for all (<= 16) queries: it also zeroes the candidate counters and the flag of this call (no memset launch).
    __shared__ float part[4];
    bool bad = false;
    for (int b = 0; b < nb; ++b) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < dim; c += 256) {
            const float v = queries[(int64_t)b * dim + c];
            ss = fmaf(v, v, ss);
        }
        ss = wave_sum(ss);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float qn = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
            const float m = mode == SCAN_COSINE ? m_rel : m_rel * e_norm_bound * qn;
            const float t = topk[(int64_t)b * k + (k - 1)] - 2.0f * m;
            thr[b] = t;
            cnt[b] = 0u;
            bad |= !(t > -INFINITY);  // NaN or -inf
        }
    }
    if (threadIdx.x == 0) *flag = bad ? 1u : 0u;
}

// Rows whose approximate score reaches thr[b] -> ids[b * cap + p] (any order; with row_norm their norms next to them, for the
// cosine transform of the re-scored candidates), cnt[b] = how many; more than cap sets *flag.
__global__ __launch_bounds__(256) void collect_above_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, const float* __restrict__ thr,
                                                             const float* __restrict__ row_norm, int32_t cap, int32_t* __restrict__ ids,
                                                             float* __restrict__ norms, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag) {
    const int b = blockIdx.y;
    const float t = thr[b];
    const float* s = scores + (int64_t)b * ld;
    auto visit = [&](float v, int64_t i) {
        if (v >= t) {
            const uint32_t p = atomicAdd(cnt + b, 1u);
            if (p < (uint32_t)cap) {
                ids[(int64_t)b * cap + p] = (int32_t)i;
                if (row_norm) norms[(int64_t)b * cap + p] = row_norm[i];
            } else {
                atomicOr(flag, 1u);
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* s4 = reinterpret_cast<const f4*>(s);
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f4 v = s4[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) visit(v[u], (i << 2) + u);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) visit(s[(n4 << 2) + threadIdx.x], (n4 << 2) + threadIdx.x);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) visit(s[i], i);
    }
}

__global__ __launch_bounds__(256) void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int64_t count,
                                                          float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) dst[i] = idx[i] >= 0 ? src[idx[i]] : 0.f;
}

// dst[b * k2 + j] = src[b * ld + b * k2 + j]: query b's scores of ITS candidates out of the [nb x nb * k2] score block
__global__ __launch_bounds__(256) void diag_blocks_kernel(const float* __restrict__ src, int64_t ld, int32_t k2, int64_t count,
                                                           float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int64_t b = i / k2;
    dst[i] = src[b * ld + i];
}

}  // namespace

int launch_approx_threshold(const float* topk, int32_t nb, int32_t k, const float* queries, int32_t dim, int mode, float m_rel,
                            float e_norm_bound, float* thr, uint32_t* cnt, uint32_t* flag, hipStream_t s) {
    hipLaunchKernelGGL(approx_threshold_kernel, dim3(1), dim3(256), 0, s, topk, nb, k, queries, (int)dim, mode, m_rel, e_norm_bound, thr, cnt,
                       flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_collect_above(const float* scores, int32_t nb, int64_t n, int64_t ld, const float* thr, const float* row_norm, int32_t cap,
                         int32_t* ids, float* norms, uint32_t* cnt, uint32_t* flag, hipStream_t s) {
    if (n <= 0 || nb <= 0) return RL_OK;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((n + 4095) / 4096, 512));
    hipLaunchKernelGGL(collect_above_kernel, dim3(bx, nb), dim3(256), 0, s, scores, n, ld, thr, row_norm, cap, ids, norms, cnt, flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_gather_f32(const float* src, const int32_t* idx, int64_t count, float* dst, hipStream_t s) {
    if (count <= 0) return RL_OK;
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, idx, count, dst);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_diag_blocks(const float* src, int64_t ld, int32_t k2, int64_t count, float* dst, hipStream_t s) {
    if (count <= 0) return RL_OK;
    hipLaunchKernelGGL(diag_blocks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, ld, k2, count, dst);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
