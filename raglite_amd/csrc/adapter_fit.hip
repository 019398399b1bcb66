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

template <typename ET>
__global__ __launch_bounds__(256) void gather_rows_kernel(const ET* __restrict__ E, int dim, int64_t n_rows,
                                                           const int32_t* __restrict__ rows, int64_t n,
                                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave0; i < n; i += n_waves) {
        const int64_t r = rows[i];
        const bool ok = r >= 0 && r < n_rows;
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

int launch_gather_rows(const void* E, bool f16, int32_t dim, int64_t n_rows, const int32_t* rows, int64_t n, float* out,
                       hipStream_t s) {
    if (n <= 0) return RL_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, 256 * 16));
    if (f16)
        hipLaunchKernelGGL((gather_rows_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, static_cast<const uint16_t*>(E),
                           (int)dim, n_rows, rows, n, out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(blocks), dim3(256), 0, s, static_cast<const float*>(E),
                           (int)dim, n_rows, rows, n, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
