// 8f-4: semantic-chunking similarities (src/raglite/_split_chunks.py:54-72), the consumer of the pooled chunklet
// embeddings (a1-a3) and the producer of the cost vector of the chunk-partition MILP (which stays on the host):
//   X      = rows normalised to unit length                                        (:54-55)
//   d      = normalised mean of the normalised rows flagged `nonoutlying`           (:57-61)
//   X_mod  = X - (X . d) d, re-normalised, used unless some ||X_mod row|| <= eps    (:62-65)
//   sim[i] = max((X[i] . X[i+1] + 1) / 2, sqrt(eps))   for consecutive rows of one document (:68-72)
// Batched over documents (doc_offsets CSR over the concatenated rows); fp32 like the reference.  Three small
// HBM-bound kernels: row norms -> per-document discourse vector -> per-row-pair similarities (+ the per-document
// "degenerate" flag) -> choice.  sim[last row of a document] = 0 (unused).
#include "common.h"

namespace rl {
namespace {

constexpr float EPS32 = 1.1920928955078125e-07f;  // np.finfo(np.float32).eps

__device__ __forceinline__ int doc_of(const int64_t* __restrict__ off, int64_t n_docs, int64_t row) {
    int64_t lo = 0, hi = n_docs;  // off[lo] <= row < off[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (off[mid] <= row) lo = mid; else hi = mid;
    }
    return (int)lo;
}

// inv_norm[i] = 1 / ||x_i||   (one wave per row)
__global__ __launch_bounds__(256) void ps_norms_kernel(const float* __restrict__ X, int64_t n, int dim,
                                                        float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += n_waves) {
        float ss = 0.f;
        for (int k = lane; k < dim; k += 64) { const float v = X[r * (int64_t)dim + k]; ss = fmaf(v, v, ss); }
        ss = wave_sum(ss);
        if (lane == 0) inv_norm[r] = 1.0f / sqrtf(ss);
    }
}

// d[doc] = normalise(mean over selected rows of x_hat); has_d[doc] = any row selected.  One block per document,
// thread t owns columns t, t+256, ...: coalesced row reads, fixed row order (deterministic).
__global__ __launch_bounds__(256) void ps_discourse_kernel(const float* __restrict__ X, const float* __restrict__ inv_norm,
                                                            const int64_t* __restrict__ doc_off, int dim,
                                                            const uint8_t* __restrict__ sel, float* __restrict__ dvec,
                                                            int32_t* __restrict__ has_d) {
    __shared__ float part[4];
    const int doc = blockIdx.x;
    const int64_t b = doc_off[doc], e = doc_off[doc + 1];
    int cnt = 0;
    float ss = 0.f;
    for (int k0 = 0; k0 < dim; k0 += 256) {
        const int k = k0 + threadIdx.x;
        float acc = 0.f;
        cnt = 0;
        for (int64_t r = b; r < e; ++r) {
            if (sel && !sel[r]) continue;
            ++cnt;
            if (k < dim) acc += X[r * (int64_t)dim + k] * inv_norm[r];
        }
        const float mean = cnt ? acc / (float)cnt : 0.f;
        if (k < dim) { dvec[(int64_t)doc * dim + k] = mean; ss = fmaf(mean, mean, ss); }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float inv = 1.0f / sqrtf((part[0] + part[1]) + (part[2] + part[3]));
    for (int k = threadIdx.x; k < dim; k += 256) dvec[(int64_t)doc * dim + k] *= inv;
    if (threadIdx.x == 0) has_d[doc] = (sel != nullptr && cnt > 0) ? 1 : 0;
}

// For row i (one wave): plain[i] = x_hat_i . x_hat_{i+1};  mod[i] = the same on the discourse-free, re-normalised
// rows; degenerate[doc] |= ||x_mod_i|| <= eps.  NV = ceil(dim / 64) values per lane live in registers.
template <int NV>
__global__ __launch_bounds__(256) void ps_pairs_kernel(const float* __restrict__ X, const float* __restrict__ inv_norm,
                                                        const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n,
                                                        int dim, const float* __restrict__ dvec,
                                                        float* __restrict__ plain, float* __restrict__ mod,
                                                        int32_t* __restrict__ degenerate) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += n_waves) {
        const int doc = doc_of(doc_off, n_docs, r);
        const bool last = r + 1 >= doc_off[doc + 1];
        const int64_t r1 = last ? r : r + 1;
        float a[NV], b[NV], d[NV];
        const float ia = inv_norm[r], ib = inv_norm[r1];
        float pa = 0.f, pb = 0.f, dot = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int k = v * 64 + lane;
            const bool ok = k < dim;
            a[v] = ok ? X[r * (int64_t)dim + k] * ia : 0.f;
            b[v] = ok ? X[r1 * (int64_t)dim + k] * ib : 0.f;
            d[v] = ok ? dvec[(int64_t)doc * dim + k] : 0.f;
            pa = fmaf(a[v], d[v], pa);
            pb = fmaf(b[v], d[v], pb);
            dot = fmaf(a[v], b[v], dot);
        }
        pa = wave_sum(pa); pb = wave_sum(pb); dot = wave_sum(dot);
        float na = 0.f, nb = 0.f, dm = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float am = a[v] - pa * d[v], bm = b[v] - pb * d[v];
            na = fmaf(am, am, na); nb = fmaf(bm, bm, nb); dm = fmaf(am, bm, dm);
        }
        na = sqrtf(wave_sum(na)); nb = sqrtf(wave_sum(nb)); dm = wave_sum(dm);
        if (lane == 0) {
            plain[r] = last ? 0.f : dot;
            mod[r] = last ? 0.f : dm / (na * nb);
            if (na <= EPS32) atomicOr(&degenerate[doc], 1);
        }
    }
}

__global__ __launch_bounds__(256) void ps_choose_kernel(const float* __restrict__ plain, const float* __restrict__ mod,
                                                         const int64_t* __restrict__ doc_off, int64_t n_docs, int64_t n,
                                                         const int32_t* __restrict__ has_d,
                                                         const int32_t* __restrict__ degenerate, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int doc = doc_of(doc_off, n_docs, r);
    if (r + 1 >= doc_off[doc + 1]) { out[r] = 0.f; return; }
    const float s = (has_d[doc] && !degenerate[doc]) ? mod[r] : plain[r];
    out[r] = fmaxf((s + 1.0f) / 2.0f, sqrtf(EPS32));
}
}  // namespace

// scratch: float[2 * n + n + n_docs * dim] + int32[2 * n_docs]  (see partition_sim_scratch_bytes)
size_t partition_sim_scratch_bytes(int64_t n, int64_t n_docs, int32_t dim) {
    return (size_t)(3 * n + n_docs * (int64_t)dim) * sizeof(float) + (size_t)(2 * n_docs) * sizeof(int32_t) + 64;
}

int launch_partition_similarity(const float* X, int64_t n, int32_t dim, const int64_t* doc_off, int64_t n_docs,
                                const uint8_t* sel, float* out, void* scratch, hipStream_t s) {
    if (n <= 0 || n_docs <= 0) return RL_OK;
    if (dim > 4096) return RL_ERR_UNSUPPORTED;
    float* inv_norm = static_cast<float*>(scratch);
    float* plain = inv_norm + n;
    float* mod = plain + n;
    float* dvec = mod + n;
    int32_t* has_d = reinterpret_cast<int32_t*>(dvec + n_docs * (int64_t)dim);
    int32_t* degenerate = has_d + n_docs;
    RL_HIP(hipMemsetAsync(degenerate, 0, (size_t)n_docs * sizeof(int32_t), s));
    const int wblocks = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, 256 * 16));
    hipLaunchKernelGGL(ps_norms_kernel, dim3(wblocks), dim3(256), 0, s, X, n, (int)dim, inv_norm);
    hipLaunchKernelGGL(ps_discourse_kernel, dim3((unsigned)n_docs), dim3(256), 0, s, X, inv_norm, doc_off, (int)dim, sel,
                       dvec, has_d);
#define RL_PS(NV) hipLaunchKernelGGL((ps_pairs_kernel<NV>), dim3(wblocks), dim3(256), 0, s, X, inv_norm, doc_off, n_docs, n, \
                                     (int)dim, dvec, plain, mod, degenerate)
    const int nv = (dim + 63) / 64;
    if (nv <= 2) RL_PS(2); else if (nv <= 4) RL_PS(4); else if (nv <= 8) RL_PS(8); else if (nv <= 16) RL_PS(16);
    else if (nv <= 32) RL_PS(32); else RL_PS(64);
#undef RL_PS
    hipLaunchKernelGGL(ps_choose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, plain, mod, doc_off, n_docs, n,
                       has_d, degenerate, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
