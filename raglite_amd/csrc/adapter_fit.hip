// 8f-3: the device half of `update_query_adapter` (src/raglite/_query_adapter.py:153-205).  The batched vector
// search is rl_search_chunks; what is left on the device is, for every (eval, retrieved chunk), the row
//     np.argmax(chunk.embedding_matrix @ q)                                  (_query_adapter.py:174,180)
// and fetching those rows for the host's NNLS / Procrustes step (fp64 LAPACK, stays on the host).
#include "common.h"

namespace rl {
namespace {

template <typename ET>
__device__ __forceinline__ float elt(const ET* p);
template <>
__device__ __forceinline__ float elt<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float elt<uint16_t>(const uint16_t* p) {
    _Float16 h;
    __builtin_memcpy(&h, p, 2);
    return (float)h;
}

// One wave per (query, candidate chunk): fp32 dots (lane-strided fmaf chain + butterfly), first maximum on ties.
template <typename ET>
__global__ __launch_bounds__(256) void chunk_best_rows_kernel(const ET* __restrict__ E, int dim,
                                                               const float* __restrict__ Q,
                                                               const int64_t* __restrict__ offsets, int64_t n_chunks,
                                                               const int32_t* __restrict__ cand, int n_cand,
                                                               int64_t n_items, int32_t* __restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t item = wave0; item < n_items; item += n_waves) {
        const int32_t c = cand[item];
        int32_t best = -1;
        if (c >= 0 && c < n_chunks) {
            const float* q = Q + (item / n_cand) * (int64_t)dim;
            const int64_t b = offsets[c], e = offsets[c + 1];
            float best_s = -INFINITY;
            for (int64_t r = b; r < e; ++r) {
                const ET* row = E + r * (int64_t)dim;
                float s = 0.f;
                for (int k = lane; k < dim; k += 64) s = fmaf(elt<ET>(row + k), q[k], s);
                s = wave_sum(s);
                if (s > best_s || best < 0) { best_s = s; best = (int32_t)r; }  // NaN-safe first-maximum rule
            }
        }
        if (lane == 0) out_rows[item] = best;
    }
}

// Exact l2 similarity 1 - sqrt(sum (e - q)^2) of every selected (query, row) pair, and the pair's position after a
// re-sort by it.  The batched paths rank by |e|^2 + |q|^2 - 2 e.q, which is accurate except for near-duplicates
// (distance below ~1e-3, where the expansion cancels); those are exactly the hits whose reported similarity matters.
template <typename ET>
__global__ __launch_bounds__(256) void rescore_l2_kernel(const ET* __restrict__ E, int dim, const float* __restrict__ Q,
                                                          const int32_t* __restrict__ rows,
                                                          const float* __restrict__ ranked, int k, int64_t n_items,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t item = wave0; item < n_items; item += n_waves) {
        int32_t r = rows[item];
        if (ranked[item] == -INFINITY) r = -1;  // padding, or a row masked out by a filter / tombstone: stays "no hit"
        float s = 0.f;
        if (r >= 0) {
            const float* q = Q + (item / k) * (int64_t)dim;
            const ET* row = E + (int64_t)r * dim;
            for (int c = lane; c < dim; c += 64) { const float t = elt<ET>(row + c) - q[c]; s = fmaf(t, t, s); }
            s = wave_sum(s);
        }
        if (lane == 0) out[item] = r >= 0 ? 1.0f - sqrtf(s) : -INFINITY;
    }
}

// rows_out[b][j] = rows_in[b][pos[b][j]]
__global__ __launch_bounds__(256) void permute_rows_kernel(const int32_t* __restrict__ rows_in,
                                                            const int32_t* __restrict__ pos, int k, int64_t n_items,
                                                            int32_t* __restrict__ rows_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    const int32_t p = pos[i];
    rows_out[i] = p >= 0 ? rows_in[(i / k) * k + p] : -1;
}

template <typename ET>
__global__ __launch_bounds__(256) void gather_rows_kernel(const ET* __restrict__ E, int dim, int64_t n_rows,
                                                           const int32_t* __restrict__ rows, int64_t n,
                                                           float* __restrict__ out, const uint32_t* __restrict__ counts, int cap) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave0; i < n; i += n_waves) {
        if (counts && (uint32_t)(i % cap) >= counts[i / cap]) continue;  // a slot past its list's length: nothing to gather (wave-uniform)
        const int64_t r = rows[i];
        const bool ok = r >= 0 && r < n_rows;
        if constexpr (sizeof(ET) == 4) {
            if ((dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {  // 16 B per lane
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4* src = reinterpret_cast<const f4*>(E + (ok ? r : 0) * (int64_t)dim);
                f4* dst = reinterpret_cast<f4*>(out + i * (int64_t)dim);
                for (int k = lane; k < (dim >> 2); k += 64) dst[k] = ok ? src[k] : (f4){NAN, NAN, NAN, NAN};
                continue;
            }
        }
        for (int k = lane; k < dim; k += 64) out[i * (int64_t)dim + k] = ok ? elt<ET>(E + r * (int64_t)dim + k) : NAN;
    }
}
}  // namespace

int launch_chunk_best_rows(const void* E, bool f16, int32_t dim, const float* Q, const int64_t* offsets,
                           int64_t n_chunks, const int32_t* cand, int32_t n_cand, int64_t n_items, int32_t* out_rows,
                           hipStream_t s) {
    if (n_items <= 0) return RL_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_items + 3) / 4, 256 * 16));
    if (f16)
        hipLaunchKernelGGL((chunk_best_rows_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s,
                           static_cast<const uint16_t*>(E), (int)dim, Q, offsets, n_chunks, cand, (int)n_cand, n_items, out_rows);
    else
        hipLaunchKernelGGL((chunk_best_rows_kernel<float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(E),
                           (int)dim, Q, offsets, n_chunks, cand, (int)n_cand, n_items, out_rows);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_rescore_l2(const void* E, bool f16, int32_t dim, const float* Q, const int32_t* rows, const float* ranked,
                      int32_t k, int64_t n_items, float* out, hipStream_t s) {
    if (n_items <= 0) return RL_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_items + 3) / 4, 256 * 16));
    if (f16)
        hipLaunchKernelGGL((rescore_l2_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, static_cast<const uint16_t*>(E),
                           (int)dim, Q, rows, ranked, (int)k, n_items, out);
    else
        hipLaunchKernelGGL((rescore_l2_kernel<float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(E),
                           (int)dim, Q, rows, ranked, (int)k, n_items, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_permute_rows(const int32_t* rows_in, const int32_t* pos, int32_t k, int64_t n_items, int32_t* rows_out,
                        hipStream_t s) {
    if (n_items <= 0) return RL_OK;
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, s, rows_in, pos, (int)k,
                       n_items, rows_out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// rl_index_compact: dst row i = src row old_row[i], `row_bytes` bytes each (a multiple of 4); one wave per row.
__global__ __launch_bounds__(256) void compact_rows_kernel(const uint32_t* __restrict__ src, int64_t row_words,
                                                            const int64_t* __restrict__ old_row, int64_t n,
                                                            uint32_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t i = wave; i < n; i += nw) {
        const uint32_t* a = src + old_row[i] * row_words;
        uint32_t* b = dst + i * row_words;
        for (int64_t c = lane; c < row_words; c += 64) b[c] = a[c];
    }
}
int launch_compact_rows(const void* src, int64_t row_bytes, const int64_t* old_row, int64_t n, void* dst, hipStream_t s) {
    if (n <= 0) return RL_OK;
    if (row_bytes % 4) return RL_ERR_UNSUPPORTED;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, 256 * 16));
    hipLaunchKernelGGL(compact_rows_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const uint32_t*>(src), row_bytes / 4, old_row, n,
                       static_cast<uint32_t*>(dst));
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_gather_rows(const void* E, bool f16, int32_t dim, int64_t n_rows, const int32_t* rows, int64_t n, float* out,
                       hipStream_t s, const uint32_t* counts, int32_t cap) {
    if (n <= 0) return RL_OK;
    if (counts && (cap < 1 || n % cap)) return RL_ERR_INVALID;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, 256 * 16));
    if (f16)
        hipLaunchKernelGGL((gather_rows_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, static_cast<const uint16_t*>(E),
                           (int)dim, n_rows, rows, n, out, counts, (int)cap);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(E),
                           (int)dim, n_rows, rows, n, out, counts, (int)cap);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
