// Row / chunk validity masks: the filter-first branch of the reference's vector search
// (src/raglite/_search.py:105-119, `WHERE chunk_id IN (filtered chunks) ORDER BY dist LIMIT num_hits`) and the
// tombstones left by delete_documents (src/raglite/_delete.py:148-176), pushed down to the device as bitsets:
// bit i of word i/32 set <=> element i takes part.  Masked elements get score -inf before the exact top-k, and
// selected -inf slots are reported as "no hit" (id -1), exactly like the tail of a k > n search.
#include "common.h"

namespace rl {
namespace {

// row_bits[w] = bits of rows 32w .. 32w+31: chunk_bits[row_to_chunk[r]] (all ones when chunk_bits == nullptr),
// and-ed with and_rows[w] when given.
__global__ __launch_bounds__(256) void expand_chunk_bits_kernel(const uint32_t* __restrict__ chunk_bits,
                                                                 const int32_t* __restrict__ row_to_chunk,
                                                                 int64_t n_rows,
                                                                 const uint32_t* __restrict__ and_rows,
                                                                 uint32_t* __restrict__ row_bits) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one lane per row, one ballot per 64 rows
    bool on = false;
    if (r < n_rows) {
        on = true;
        if (chunk_bits) {
            const int32_t c = row_to_chunk[r];
            on = (chunk_bits[c >> 5] >> (c & 31)) & 1u;
        }
    }
    const uint64_t b = __builtin_amdgcn_ballot_w64(on);
    const int lane = threadIdx.x & 63;
    const int64_t w = r >> 5;
    if ((lane & 31) == 0 && w < ((n_rows + 31) >> 5)) {  // lanes 0 and 32 own one 32-row word each
        uint32_t word = (uint32_t)(lane ? (b >> 32) : b);
        if (and_rows) word &= and_rows[w];
        row_bits[w] = word;
    }
}

__global__ __launch_bounds__(256) void mask_scores_kernel(float* __restrict__ scores, int64_t n, int64_t ld,
                                                           const uint32_t* __restrict__ bits) {
    float* s = scores + (int64_t)blockIdx.y * ld;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t words = (n + 31) >> 5;
    // one thread per 32-element word: masked-out elements are rare or frequent, either way the word decides
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += stride) {
        uint32_t off = ~bits[w];
        const int64_t base = w << 5;
        if (base + 32 > n) off &= (n - base >= 32) ? 0xffffffffu : ((1u << (n - base)) - 1u);
        while (off) {
            const int b = __builtin_ctz(off);
            off &= off - 1;
            s[base + b] = -INFINITY;
        }
    }
}

__global__ __launch_bounds__(256) void fix_masked_kernel(const float* __restrict__ scores, int32_t* __restrict__ ids,
                                                          int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count && scores[i] == -INFINITY) ids[i] = -1;
}

__global__ __launch_bounds__(256) void popcount_kernel(const uint32_t* __restrict__ bits, int64_t n,
                                                        unsigned long long* __restrict__ out) {
    const int64_t words = (n + 31) >> 5;
    unsigned long long c = 0;
    for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < words; w += (int64_t)gridDim.x * 256) {
        uint32_t v = bits[w];
        const int64_t base = w << 5;
        if (base + 32 > n) v &= (1u << (n - base)) - 1u;
        c += __builtin_popcount(v);
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
}  // namespace

int launch_expand_chunk_bits(const uint32_t* chunk_bits, const int32_t* row_to_chunk, int64_t n_rows,
                             const uint32_t* and_rows, uint32_t* row_bits, hipStream_t s) {
    if (n_rows <= 0) return RL_OK;
    const int64_t padded = (n_rows + 63) & ~int64_t(63);
    hipLaunchKernelGGL(expand_chunk_bits_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, s, chunk_bits,
                       row_to_chunk, n_rows, and_rows, row_bits);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_mask_scores(float* scores, int32_t nb, int64_t n, int64_t ld, const uint32_t* bits, hipStream_t s) {
    if (n <= 0 || nb <= 0) return RL_OK;
    const int64_t words = (n + 31) >> 5;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((words + 255) / 256, 1024));
    hipLaunchKernelGGL(mask_scores_kernel, dim3(bx, nb), dim3(256), 0, s, scores, n, ld, bits);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_fix_masked(const float* scores, int32_t* ids, int64_t count, hipStream_t s) {
    if (count <= 0) return RL_OK;
    hipLaunchKernelGGL(fix_masked_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, scores, ids, count);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_popcount(const uint32_t* bits, int64_t n, unsigned long long* out_dev, hipStream_t s) {
    RL_HIP(hipMemsetAsync(out_dev, 0, sizeof(unsigned long long), s));
    if (n <= 0) return RL_OK;
    const int64_t words = (n + 31) >> 5;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((words + 255) / 256, 512));
    hipLaunchKernelGGL(popcount_kernel, dim3(bx), dim3(256), 0, s, bits, n, out_dev);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
