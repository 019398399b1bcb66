// a9 at batch scale: MaxSim of EIGHT queries (up to 32 vectors each) per corpus pass over the PRE-SPLIT corpus image.
//
//   out[q * out_stride + c] = sum_{i < nq} max_{j in chunk c} Q_q[i] . D[j]      q = 0 .. n_q - 1  (n_q <= 8)
//
// Multi-query generalisation of src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174, behind the reranker
// plugin call src/raglite/_search.py:394-396 (rl_maxsim_topk_batch).
//
// Why a third MaxSim kernel.  The streaming kernels of maxsim_stream.hip keep the query in registers and the corpus
// tile in LDS: one (or two) queries per corpus pass, HBM-bound at 0.58-0.68 ms per pass while the fp16 matrix pipe idles
// (27 % busy, profiles/r01_m_pmc.txt) and every fp32 corpus element is re-split into its fp16 (hi, lo) pair on every
// pass.  Per query the split arithmetic needs 3 x 2 x 32 x N x 1024 flop = 0.079 ms of the 2.5 PFLOP/s fp16 pipe at
// N = 1 M, so the matrix roof is 4x away.  This kernel is the GEMM formulation of the same sum:
//   * the corpus is split ONCE, when the index is built (presplit_rows_kernel): x * e_scale = hi + lo, hi = fp16_rtz, lo =
//     fp16_rtz(x * e_scale - hi), stored as the kernel's own LDS image -- blocks of 16 rows, per block and 32-wide K slab
//     2 KiB holding the rows' 4 hi chunks and 4 lo chunks of 16 B, already swizzled (chunk c of row j at position
//     c ^ ((j >> 1) & 7)) so that every LDS-DMA instruction copies 1 KiB of contiguous HBM and every ds_read_b128
//     fragment read is bank-conflict-free.  4 B per element: the same HBM bytes per pass as the fp32 matrix;
//   * tile = 256 corpus rows x 256 query vectors (8 queries x 32), K slabs of 32: one persistent 512-thread workgroup
//     per CU walks a contiguous, chunk-aligned row range; wave w owns query w: its accumulators are the 32 x 256 score
//     tile S^T[query vector][corpus row] (2 x 16 MFMA tiles = 128 VGPRs), summed over all of K, so there is no K-split
//     reduction and no partial exchange;
//   * the corpus slab (32 KiB) goes HBM -> LDS by `global_load_lds_dwordx4 ... nt`, every wave issuing 4 of the 32 DMAs
//     from inside its MFMA loop, ring of 3 slabs, ONE workgroup barrier per slab; the query fragments (4 KiB per wave
//     and slab, L2-resident: the whole batch's image is 1 MiB) go straight from L2 into registers, one slab ahead;
//   * per slab and wave 96 v_mfma_f32_16x16x32_f16 (q_hi.e_hi, q_hi.e_lo, q_lo.e_hi -- the third skipped when the
//     query's lo halves are all zero) on 36 ds_read_b128: zero conversion VALU in the loop;
//   * the MFMA operands are swapped against maxsim_stream.hip (A = query fragment, B = corpus fragment), so a lane of
//     the C/D layout holds 4 query vectors x ONE corpus row and the 16 corpus rows of a block lie along a DPP row: the
//     per-chunk maximum is a branch-free segmented max-scan over 16 lanes (row_shr 1, 2, 4, 8 under scalar lane masks
//     derived from a per-row "last row of its chunk" bitmap), the open chunk's running maxima move to the next block by
//     row_ror, the sum over the 32 query vectors is 7 adds + 2 cross-row shuffles, and a finished chunk is stored by
//     its end row's lane.  No LDS round trip, no per-row branch.
// Deterministic: fixed MFMA order per (query vector, row), fixed scan and sum order; results do not depend on the grid.
// Integer-valued data is exact (products of fp16 halves are exact in fp32): bit-identical to the oracle.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int MG_TM = 256;               // corpus rows per tile
constexpr int MG_NBLK = MG_TM / 16;      // 16-row blocks per tile
constexpr int MG_SLAB = MG_TM * 128;     // bytes of one K slab (32 k) of a tile in LDS: 16 blocks x 2 KiB
constexpr int MG_NSLOT = 4;              // LDS ring: slab g being multiplied, g + 1 landed, g + 2 and g + 3 in flight
constexpr int MG_WAVES = 8;              // waves per workgroup = queries per pass

__device__ __forceinline__ int64_t mg_uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
}  // namespace

// ---- corpus image ----------------------------------------------------------------------------------------------------
// Bytes of the image of `rows` rows (rounded up to whole 16-row blocks).
size_t planes_bytes(int64_t rows, int32_t dim, bool half) { return (size_t)((rows + 15) / 16) * 16 * (size_t)dim * (half ? 2 : 4); }

// One thread per (row, slab, k-quarter): 8 consecutive fp32 -> one hi chunk and one lo chunk.  Rows in
// [n_rows, 16 * ceil(n_rows / 16)) are written as zeros.  `first_row` must be a multiple of 16 unless the rows before
// it in its block are already in place (append).
__global__ __launch_bounds__(256) void presplit_rows_kernel(const float* __restrict__ E, int64_t first_row, int64_t end_row,
                                                             int64_t n_rows, int32_t dim, float scale, char* __restrict__ planes) {
    const int32_t nslab = dim >> 5;
    const int64_t per_row = (int64_t)nslab * 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = first_row + i / per_row;
    if (row >= end_row) return;
    const int32_t rem = (int32_t)(i % per_row), s = rem >> 2, kq = rem & 3;
    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < n_rows) {
        const float* p = E + row * dim + 32 * s + 8 * kq;
        v0 = *reinterpret_cast<const f32x4*>(p);
        v1 = *reinterpret_cast<const f32x4*>(p + 4);
    }
    h16x8 hi, lo;
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
        const float x0 = (u < 4 ? v0[u] : v1[u - 4]) * scale, x1 = (u < 4 ? v0[u + 1] : v1[u - 3]) * scale;
        const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);  // truncation: the residual is exact in fp32
        const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
        hi[u] = ph[0]; hi[u + 1] = ph[1];
        lo[u] = pl[0]; lo[u + 1] = pl[1];
    }
    const int j = (int)(row & 15), sw = (j >> 1) & 7;
    char* blk = planes + (((row >> 4) * nslab + s) * 16 + j) * 128;
    *reinterpret_cast<h16x8*>(blk + ((kq ^ sw) << 4)) = hi;
    *reinterpret_cast<h16x8*>(blk + (((4 + kq) ^ sw) << 4)) = lo;
}

int launch_presplit_rows(const float* E, int64_t first_row, int64_t n_rows, int32_t dim, float scale, void* planes, hipStream_t s) {
    if (dim % 32 || dim < 32) return RL_ERR_UNSUPPORTED;
    const int64_t end_row = (n_rows + 15) / 16 * 16;
    if (end_row <= first_row) return RL_OK;
    const int64_t threads = (end_row - first_row) * (dim / 8);
    hipLaunchKernelGGL(presplit_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, E, first_row, end_row, n_rows, dim,
                       scale, static_cast<char*>(planes));
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// fp16-stored corpus -> its one-plane image (a pure permutation, 2 B per element): the 1 KiB of a (16-row block, slab) holds
// k-chunk kq of row j at 256 kq + 16 j, i.e. in the order the 64 lanes of a wave read it (and a 1-KiB LDS-DMA writes it).
__global__ __launch_bounds__(256) void preformat_rows16_kernel(const uint16_t* __restrict__ E, int64_t first_row, int64_t end_row,
                                                                int64_t n_rows, int32_t dim, char* __restrict__ planes) {
    const int32_t nslab = dim >> 5;
    const int64_t per_row = (int64_t)nslab * 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = first_row + i / per_row;
    if (row >= end_row) return;
    const int32_t rem = (int32_t)(i % per_row), s = rem >> 2, kq = rem & 3;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < n_rows) v = *reinterpret_cast<const uint4*>(E + row * dim + 32 * s + 8 * kq);
    *reinterpret_cast<uint4*>(planes + ((row >> 4) * nslab + s) * 1024 + kq * 256 + (row & 15) * 16) = v;
}

// The HI halves of the fp16 split of an fp32 corpus as a one-plane image (the layout of preformat_rows16_kernel): what the
// approximate MaxSim pass of big batches multiplies (api.hip: maxsim_batch_hi) -- fp16(e * scale) rounded to NEAREST even: what the
// halves drop is then at most half an ulp (max_row_norm_kernel measures it with the same rounding; toward zero, the first version,
// left 40 % more candidates for the exact re-scoring: 418 instead of 301 per query on the benchmark corpus).
__global__ __launch_bounds__(256) void presplit_hi_rows_kernel(const float* __restrict__ E, int64_t first_row, int64_t end_row,
                                                                int64_t n_rows, int32_t dim, float scale, char* __restrict__ planes) {
    const int32_t nslab = dim >> 5;
    const int64_t per_row = (int64_t)nslab * 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = first_row + i / per_row;
    if (row >= end_row) return;
    const int32_t rem = (int32_t)(i % per_row), s = rem >> 2, kq = rem & 3;
    f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < n_rows) {
        const float* p = E + row * dim + 32 * s + 8 * kq;
        v0 = *reinterpret_cast<const f32x4*>(p);
        v1 = *reinterpret_cast<const f32x4*>(p + 4);
    }
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    auto pk = [&](float a, float b) -> uint32_t {
        uint32_t w;
        const h16x2 t = (h16x2){(_Float16)(a * scale), (_Float16)(b * scale)};
        __builtin_memcpy(&w, &t, 4);
        return w;
    };
    const uint4 o = make_uint4(pk(v0[0], v0[1]), pk(v0[2], v0[3]), pk(v1[0], v1[1]), pk(v1[2], v1[3]));
    *reinterpret_cast<uint4*>(planes + ((row >> 4) * nslab + s) * 1024 + kq * 256 + (row & 15) * 16) = o;
}

int launch_presplit_hi_rows(const float* E, int64_t first_row, int64_t n_rows, int32_t dim, float scale, void* planes, hipStream_t s) {
    if (dim % 32 || dim < 32 || (reinterpret_cast<uintptr_t>(E) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t end_row = (n_rows + 15) / 16 * 16;
    if (end_row <= first_row) return RL_OK;
    const int64_t threads = (end_row - first_row) * (dim / 8);
    hipLaunchKernelGGL(presplit_hi_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, E, first_row, end_row, n_rows, dim,
                       scale, static_cast<char*>(planes));
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_preformat_rows16(const uint16_t* E, int64_t first_row, int64_t n_rows, int32_t dim, void* planes, hipStream_t s) {
    if (dim % 32 || dim < 32 || (reinterpret_cast<uintptr_t>(E) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t end_row = (n_rows + 15) / 16 * 16;
    if (end_row <= first_row) return RL_OK;
    const int64_t threads = (end_row - first_row) * (dim / 8);
    hipLaunchKernelGGL(preformat_rows16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, E, first_row, end_row, n_rows, dim,
                       static_cast<char*>(planes));
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ends[w] bit i <=> row 32 w + i is the last row of its chunk (row_to_chunk[n_rows] = -1 terminates the last chunk).
// `words` words are written; rows >= n_rows give 0.
__global__ __launch_bounds__(256) void chunk_ends_kernel(const int32_t* __restrict__ row_to_chunk, int64_t n_rows, int64_t words,
                                                          uint32_t* __restrict__ ends) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint32_t m = 0;
    const int64_t r0 = w * 32;
    if (r0 < n_rows) {
        int32_t c = row_to_chunk[r0];
        for (int i = 0; i < 32 && r0 + i < n_rows; ++i) {
            const int32_t nx = row_to_chunk[r0 + i + 1];
            m |= (uint32_t)(nx != c) << i;
            c = nx;
        }
    }
    ends[w] = m;
}

size_t chunk_ends_words(int64_t rows) { return (size_t)((rows + 31) / 32) + 16; }  // + window over-read of the last tile

int launch_chunk_ends(const int32_t* row_to_chunk, int64_t n_rows, uint32_t* ends, hipStream_t s) {
    const int64_t words = (int64_t)chunk_ends_words(n_rows);
    hipLaunchKernelGGL(chunk_ends_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, row_to_chunk, n_rows, words, ends);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- query image -----------------------------------------------------------------------------------------------------
// One block per query: scale (one power of two per query, the largest |element| to [2^13, 2^14)), split into fp16
// (hi, lo) and lay the MFMA A fragments out exactly as a wave loads them:
//   frag[((query * nslab + s) * 4 + 2 * qb + part) * 64 + lane]   16 B = 8 halves, part 0 = hi, 1 = lo;
//   lane (j = lane & 15, kq = lane >> 4) holds k = 32 s + 8 kq .. + 7 of query vector 16 qb + j (zeros past nq)
//   meta[2 * query + {0, 1}] = {2^(ex - 14) (undoes the scale), any lo != 0}
//   qsum[2 * query + {0, 1}] = {sum_i |q_i|, sum_i |q_lo,i|} with q_lo = q - fp16(q * scale) / scale: what the error bound of the one-product
//   pass is made of (hi_filter.hip: maxsim_threshold_kernel's statements, a wave per query vector -- the same bits) -- the batch pipeline
//   then needs no threshold kernel that reads the queries a third time
//   zero_words[0 .. n_zero): zeroed by query 0's block (the batch's flag words: no memset launch)
__global__ __launch_bounds__(1024) void query_planes_kernel(const float* __restrict__ Q, int nq, int dim, int64_t q_stride,
                                                            uint4* __restrict__ frag, float* __restrict__ meta, float* __restrict__ qsum,
                                                            uint32_t* __restrict__ zero_words, int n_zero) {
    __shared__ float part[16];  // (1024 threads: with 256 the two walks over the query were 26 us of latency per 128-query step)
    __shared__ float part_n[16], part_lo[16];
    __shared__ int any_lo_sh;
    if (blockIdx.x == 0 && (int)threadIdx.x < n_zero) zero_words[threadIdx.x] = 0u;
    const int64_t query = blockIdx.x;
    const float* const Qg = Q + query * q_stride;
    const int nslab = dim >> 5;
    float mx = 0.f;
    for (int i = threadIdx.x * 4; i < nq * dim; i += 4096) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Qg + i);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    if (threadIdx.x == 0) any_lo_sh = 0;
    __syncthreads();
    mx = part[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, part[i]);
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
    ex = ex > -100 ? ex : -100;
    const float q_scale = ldexpf(1.f, 14 - ex);
    bool any_lo = false;
    uint4* const out = frag + query * nslab * 4 * 64;
    for (int t = threadIdx.x; t < nslab * 2 * 64; t += 1024) {
        const int lane = t & 63, qb = (t >> 6) & 1, s = t >> 7;
        const int qi = 16 * qb + (lane & 15), kq = lane >> 4;
        h16x8 hi8, lo8;
        f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (qi < nq) {
            const float* p = Qg + (int64_t)qi * dim + 32 * s + 8 * kq;
            v0 = *reinterpret_cast<const f32x4*>(p);
            v1 = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = (u < 4 ? v0[u] : v1[u - 4]) * q_scale;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)hi);
            hi8[u] = hi;
            lo8[u] = lo;
            any_lo |= lo != (_Float16)0.0f;
        }
        uint4 a, b;
        __builtin_memcpy(&a, &hi8, 16);
        __builtin_memcpy(&b, &lo8, 16);
        out[(s * 4 + 2 * qb + 0) * 64 + lane] = a;
        out[(s * 4 + 2 * qb + 1) * 64 + lane] = b;
    }
    if (any_lo) any_lo_sh = 1;  // benign race: every writer stores 1
    // sum_i |q_i| and sum_i |q_lo,i|: wave w takes the vectors w, w + 16 (maxsim_threshold_kernel: lane l sums the elements 4 (l + 64 j) .. + 3)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const float inv_scale = ldexpf(1.f, ex - 14);
        float w_norms = 0.f, w_lo = 0.f;
        const bool vec = (dim & 3) == 0;  // (dim % 32 == 0 here; Q and q_stride are 16-byte aligned: launch_query_planes)
        for (int i = wv; i < nq; i += 16) {
            float ss = 0.f, sl = 0.f;
            auto add = [&](float v) {
                ss = fmaf(v, v, ss);
                const float x = v * q_scale;
                const float lo = x - (float)(_Float16)x;
                sl = fmaf(lo, lo, sl);
            };
            if (vec) {
                const f32x4* row = reinterpret_cast<const f32x4*>(Qg + (int64_t)i * dim);
                for (int c = lane; c < (dim >> 2); c += 64) {
                    const f32x4 a = row[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) add(a[u]);
                }
            }
            w_norms += sqrtf(wave_sum(ss));
            w_lo += sqrtf(wave_sum(sl)) * inv_scale;
        }
        if (lane == 0) { part_n[wv] = w_norms; part_lo[wv] = w_lo; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        meta[2 * query + 0] = ldexpf(1.f, ex - 14);
        meta[2 * query + 1] = any_lo_sh ? 1.f : 0.f;
        float sn = 0.f, sl = 0.f;
        for (int i = 0; i < 16; ++i) { sn += part_n[i]; sl += part_lo[i]; }
        qsum[2 * query + 0] = sn;
        qsum[2 * query + 1] = sl;
    }
}

size_t query_planes_bytes(int32_t dim, int32_t n_queries) { return (size_t)n_queries * ((size_t)dim * 128 + 16) + 64; }

int launch_query_planes(const float* Q, int32_t dim, int32_t nq, int64_t q_stride, int32_t n_queries, void* buf, hipStream_t s, uint32_t* zero_words,
                        int n_zero) {
    if (n_zero < 0 || n_zero > 1024 || (n_zero > 0 && !zero_words)) return RL_ERR_INVALID;
    if (nq < 1 || nq > 32 || n_queries < 1 || dim % 32 || dim < 32) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Q) & 15) || (q_stride & 3)) return RL_ERR_UNSUPPORTED;
    uint4* frag = static_cast<uint4*>(buf);
    float* meta = reinterpret_cast<float*>(static_cast<char*>(buf) + (size_t)n_queries * dim * 128);
    float* qsum = meta + 2 * (size_t)n_queries;  // (query_planes_qsum)
    hipLaunchKernelGGL(query_planes_kernel, dim3((unsigned)n_queries), dim3(1024), 0, s, Q, (int)nq, (int)dim, q_stride, frag, meta, qsum, zero_words, n_zero);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- the pass --------------------------------------------------------------------------------------------------------
namespace {
// One query fragment load (64 lanes x 16 B, L2-resident) into 4 VGPRs, invisible to the compiler's vmcnt bookkeeping:
// the value is only valid after the matching `s_waitcnt vmcnt` + mg_pin() below.
__device__ __forceinline__ void mg_load_frag(f32x4& dst, uint32_t voff, const char* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void mg_pin(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

}  // namespace

// dst = max(src shifted along its DPP row, x); lanes without a source lane keep an undefined dst (never selected below).
// `volatile`: the epilogue issues these in source order -- all DPP reads of a step, then all selects -- so that a VGPR
// written by a select is read through DPP at least 8 instructions later (the VALU-write -> DPP-read hazard needs 2 wait
// states and the compiler's hazard recogniser does not look inside inline asm).
#define MG_MAX_DPP(dst, src, x, CTRLSTR) asm volatile("v_max_f32_dpp %0, %1, %2 " CTRLSTR " row_mask:0xf bank_mask:0xf" : "=&v"(dst) : "v"(src), "v"(x))
#define MG_SELECT(x, t, mask) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(t), "s"(mask))

// MODE 1: the same main loop as a row-score GEMM (a6 for many queries, BASELINE cfg 5): the 8 waves of a tile score 8 groups
// of 32 independent queries, the metric of src/raglite/_typing.py:123-134 is applied in the epilogue and S[q * ld + row] is
// stored; a workgroup walks (row tile, query tile) pairs, all query tiles of a row tile back to back.
struct RowScoreArgs {
    float* S; int64_t ld;                 // [B x ld] similarities
    const float* row_norm;                // cosine: |e| per row;  l2: |e|^2 per row
    const float* q_sumsq;                 // |q|^2 per query (cosine / l2)
    const float* q_unscale;               // per query: 2^(ex - 14), undoes its scale
    const float* q_anylo;                 // per group of 32 queries: any lo half != 0
    int32_t B, QT, metric;                // queries, query tiles of 256 per row tile, SCAN_* mode
    int32_t tile_stride;                  // only every tile_stride-th 256-row tile is scored (1 = all; > 1: the sample pass)
    int32_t compact;                      // MODE 1 with a stride: column of S = sampled tile ordinal * 256 + row in tile
    int q_outer;                          // walk order of the (row tile, query tile) pairs, see the kernel
    const uint32_t* run_if;               // the kernel returns at once unless *run_if != 0 (nullptr: always runs)
    // MODE 2 (fused exact top-k, no score matrix): rows whose similarity reaches tau[q] are appended to per-query lists
    const float* tau; int32_t tau_stride; // threshold of query q: tau[q * tau_stride]
    float* cand_scores; int32_t* cand_ids;  // [B x cap], pre-filled with (-inf, -1)
    uint32_t* cand_cnt;                   // [B] zeroed
    uint32_t* overflow;                   // set when a list overflows or a threshold is unusable: the caller's dense path then runs
    int32_t cap;
};

// HALF: the image of an fp16-STORED corpus (rl_index_create_f16; the reference's pgvector halfvec column,
// src/raglite/_typing.py:211-232): one fp16 plane, 2 B per element -- a (block, slab) is 1 KiB laid out exactly as a wave
// reads it (lane 16 kq + j <- k-chunk kq of row j), the slab 16 KiB, and each product is 2 MFMAs (q_hi.e + q_lo.e), exact
// for the stored values.
// HO (HALF images, one product -- what every one-product launch runs since round 3: cfg 5's fused top-k over the HI image 4.18 -> 3.74 ms,
// the eight-query MaxSim pass 0.71 -> 0.65 ms, profiles/r03_a / r03_f): the pass with a DEEP corpus stream.  Measured
// (profiles/r02_u_skeleton.txt): without any MFMA the HALF pass still takes 0.51 ms -- the image moves at 4 TB/s -- and the matrix work
// is added to that, not hidden by it.  Cause: VMEM retires in order and a wave's query-fragment loads of slab g are issued one
// slab ahead, i.e. after every DMA but the newest slab's: waiting for them at the top of slab g drains all older DMAs -- one 16-KiB
// slab per CU in flight, however deep the ring.  Here the
// query fragments (hi halves only: two loads per slab) are loaded THREE slabs ahead into four register sets and the ring has six
// slots: at the top of slab g the DMAs of slabs g + 2 .. g + 4 stay in flight (48 KiB per CU).
template <int NQB, bool TRACE = false, int MODE = 0, bool HALF = false, bool HO = false>
__global__ __launch_bounds__(512, 2) void maxsim_gemm_kernel(const char* __restrict__ planes, int64_t n_rows, int32_t nslab,
                                                               const char* __restrict__ qfrag, const float* __restrict__ qmeta,
                                                               int32_t n_q, const int32_t* __restrict__ row_to_chunk,
                                                               const int64_t* __restrict__ chunk_offsets,
                                                               const uint32_t* __restrict__ ends_bits, float* __restrict__ out,
                                                               int64_t out_stride, float inv_e_scale, int dbg, unsigned long long* trace,
                                                               RowScoreArgs rs) {
    // TRACE (diagnostic build, RAGLITE_GEMM_TRACE=1): s_memtime stamps of workgroup 7, slabs 128..143, kept in
    // LDS and copied out at the end: [slab - 128][wave][stamp 0..15] (3 + p = after pair p of the slab).
    static_assert(!HO || (HALF && !TRACE), "HO: the one-product pass over a one-plane image");
    constexpr int BLKB = HALF ? 1024 : 2048;   // bytes of one (16-row block, K slab) of the image
    constexpr int SLAB = MG_NBLK * BLKB;       // one K slab of a tile in LDS
    constexpr int NSLOT = HO ? 6 : MG_NSLOT;   // LDS ring
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLAB + (TRACE ? 16 * 8 * 16 * 8 : 0) + (MODE == 2 ? MG_WAVES * 4096 : 0)];
    if (rs.run_if && __builtin_amdgcn_readfirstlane((int)*rs.run_if) == 0) return;  // whole grid: the guarded fallback is not needed
    auto stamp = [&](int g, int k) {
        if constexpr (TRACE) {
            if (blockIdx.x == 7 && g >= 128 && g < 144 && (threadIdx.x & 63) == 0)
                reinterpret_cast<unsigned long long*>(smem + NSLOT * SLAB)[((g - 128) * 8 + (threadIdx.x >> 6)) * 16 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    if constexpr (TRACE) {
        for (int i = threadIdx.x; i < 16 * 8 * 16; i += blockDim.x) reinterpret_cast<unsigned long long*>(smem + NSLOT * SLAB)[i] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();  // 0..7 = the query this wave scores
    const int64_t G = gridDim.x, b = blockIdx.x;
    if constexpr (MODE == 0) {  // gridDim.y > 1: ONE launch for several passes (the guarded fallback of a whole batch): pass blockIdx.y scores
                                // queries 8 y .. 8 y + 7 of the n_q the launch was given -- its own fragments, meta words and output rows
        const int y = (int)blockIdx.y;
        qfrag += (int64_t)y * MG_WAVES * nslab * 4096;
        qmeta += 2 * MG_WAVES * y;
        out += (int64_t)y * MG_WAVES * out_stride;
        n_q = n_q - MG_WAVES * y < MG_WAVES ? n_q - MG_WAVES * y : MG_WAVES;
    }
    // Chunk-aligned row range of this workgroup (as in maxsim_stream.hip): first chunk boundary at or after n_rows * b / G.
    auto boundary = [&](int64_t t) -> int64_t {
        if (t <= 0) return 0;
        if (t >= n_rows) return n_rows;
        const int32_t c = row_to_chunk[t];
        const int64_t c0 = chunk_offsets[c], c1 = chunk_offsets[c + 1];
        return c0 == t ? t : c1;
    };
    int32_t r_lo = 0, r_hi = 0, vt0 = 0, vt1 = 0, Tv = 1;  // MODE 1/2: pairs [vt0, vt1) of Tv row tiles x QT query tiles
    const int32_t stride = MODE == 0 ? 1 : rs.tile_stride;
    if constexpr (MODE == 0) {
        r_lo = (int32_t)mg_uniform_i64(boundary((n_rows * b) / G));
        r_hi = (int32_t)mg_uniform_i64((b + 1 == G) ? n_rows : boundary((n_rows * (b + 1)) / G));
        if (r_hi <= r_lo) return;  // whole workgroup
    } else {  // whole 256-row tiles, no chunk structure; with a stride only the tiles 0, stride, 2 stride, ...
        // The workgroups share the (row tile, query tile) PAIRS evenly, not the row tiles: 4 883 row tiles over 256 workgroups
        // are 19 or 20 each and the 20s would set the pace (+ 4.9 %); 19 532 pairs are 76 or 77 each.
        const int64_t T = (n_rows + MG_TM - 1) / MG_TM;
        Tv = (int32_t)((T + stride - 1) / stride);
        const int64_t items = (int64_t)Tv * rs.QT;
        vt0 = (int32_t)((items * b) / G);
        vt1 = (int32_t)((items * (b + 1)) / G);
        if (vt1 <= vt0) return;  // whole workgroup
        r_hi = (int32_t)n_rows;
    }
    const int32_t org = r_lo & ~15;                       // MODE 0: tiles start on a 16-row block of the image
    const int QT = MODE == 0 ? 1 : rs.QT;                 // query tiles per row tile
    const int nt = MODE == 0 ? (r_hi - org + MG_TM - 1) / MG_TM : vt1 - vt0;  // tiles (MODE 1/2: pairs) of this workgroup
    // MODE 1/2 order of the pairs: query tile OUTERMOST (pair i = query tile i / Tv, row tile i % Tv) -- at any time the whole
    // chip works on one 1 MiB query-tile image, as in MODE 0; q_outer = 0 (query tile fastest: each corpus tile QT times back
    // to back) measured the same on cfg 5 (8.62 vs 8.73 ms): neither order is memory-limited.
    const bool q_outer = MODE != 0 && rs.q_outer != 0;
    auto tile_row0 = [&](int t) -> int32_t { return MODE == 0 ? org + t * MG_TM : (q_outer ? (vt0 + t) % Tv : (vt0 + t) / QT) * stride * MG_TM; };
    const int total = nt * nslab;                         // K slabs this workgroup consumes, tile after tile
    const int32_t last_blk = (int32_t)((n_rows + 15) >> 4) - 1;
    // MODE 0: wave = query wv of the pass.  MODE 1: wave = group (tile % QT) * 8 + wv of 32 queries; n_q counts the groups.
    auto group_of = [&](int t) { return MODE == 0 ? wv : (q_outer ? (vt0 + t) / Tv : (vt0 + t) % QT) * MG_WAVES + wv; };
    bool has_q = group_of(0) < n_q;                       // wave-uniform; per tile in MODE 1
    const int fj = lane & 15, kq = lane >> 4;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;

    // ---- the corpus stream: waves 0-3 bring blocks 4 wv .. 4 wv + 3 of every slab (8 DMAs of 1 KiB each) ----------------
    // Measured (profiles/r02_c_trace.txt): of the two waves of a SIMD the first-dispatched one (waves 0-3) wins the
    // arbitration for the matrix pipe and finishes its 96 MFMAs of a slab ~1.5 k cycles before its partner, then waits at
    // the barrier.  So the older waves carry the whole DMA duty (an LDS-DMA instruction stalls its wave for 60-185
    // cycles): it costs them slack, not the critical path.
    const bool feeder = (dbg & 16) ? wv >= 4 : wv < 4;  // wave-uniform
    struct Feed { const char* src[4]; uint32_t lds; };
    int f_tile = 0, f_s = 0, f_slot = 0;   // position of the NEXT slab to fetch
    const char* f_base[4];                 // block bases of the tile being fetched
    auto feed_tile = [&](int t) __attribute__((always_inline)) {
        const int32_t b0 = (tile_row0(t) >> 4) + 4 * (wv & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int32_t blk = b0 + i;
            blk = blk < last_blk ? blk : last_blk;  // past the image: harmless re-read of the last block, never used
            f_base[i] = planes + mg_uniform_i64((int64_t)blk * nslab * BLKB);
        }
    };
    feed_tile(0);
    auto next_feed = [&]() __attribute__((always_inline)) {
        Feed f;
#pragma unroll
        for (int i = 0; i < 4; ++i) f.src[i] = f_base[i] + f_s * BLKB;
        f.lds = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(f_slot * SLAB + 4 * (wv & 3) * BLKB));
        if (++f_s == nslab) {  // past the end: re-fetch the last tile (keeps the vmcnt bookkeeping uniform)
            f_s = 0;
            if (f_tile + 1 < nt) { ++f_tile; feed_tile(f_tile); }
        }
        f_slot = f_slot + 1 == NSLOT ? 0 : f_slot + 1;
        return f;
    };
    auto dma_piece = [&](const Feed& f, auto P_) {  // piece p = 2 * block + half (HALF: a block is one piece)
        constexpr int p = decltype(P_)::value, i = p >> 1, h = p & 1;
        if constexpr (!(HALF && h == 1)) {
            const uint32_t lane16 = 16u * lane;  // (asm operands alone do not capture an enclosing local in a generic lambda)
            const char* const src = reinterpret_cast<const char*>(mg_uniform_i64(reinterpret_cast<int64_t>(f.src[i])));
            const uint32_t lds = __builtin_amdgcn_readfirstlane(f.lds);
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:%4 nt" ::"s"(lds), "n"(i * BLKB),
                         "v"(lane16), "s"(src), "n"(h * 1024)
                         : "memory", "m0", "scc");
        }
    };
    auto dma_all = [&](const Feed& f) __attribute__((always_inline)) {
        [&]<int... P>(std::integer_sequence<int, P...>) { (dma_piece(f, std::integral_constant<int, P>{}), ...); }
        (std::make_integer_sequence<int, 8>{});
    };

    // ---- this wave's query fragments: slab s -> 4 x 16 B per lane (qb0 hi, qb0 lo, qb1 hi, qb1 lo) -------------------
    auto qbase_of = [&](int t) {  // (clamped: a wave without a group loads some valid fragments and ignores them)
        const int grp = group_of(t) < n_q ? group_of(t) : n_q - 1;
        return qfrag + (int64_t)grp * nslab * 4096;
    };
    const char* qbase = qbase_of(0);
    int q_s = 0, q_tile = 0;  // slab and tile of the NEXT fragment load
    auto q_advance = [&]() __attribute__((always_inline)) {
        if (++q_s == nslab) {
            q_s = 0;
            if constexpr (MODE != 0) { q_tile = q_tile + 1 < nt ? q_tile + 1 : q_tile; qbase = qbase_of(q_tile); }
        }
    };
    auto load_q = [&](f32x4 (&q)[4]) __attribute__((always_inline)) {
        const char* p = qbase + (int64_t)q_s * 4096;
        mg_load_frag(q[0], lane16, p);
        mg_load_frag(q[1], lane16, p + 1024);
        mg_load_frag(q[2], lane16, p + 2048);
        mg_load_frag(q[3], lane16, p + 3072);
        q_advance();
    };
    const float unscale = (MODE == 0 && has_q) ? qmeta[2 * wv] * inv_e_scale : 0.f;
    auto any_lo_of = [&](int t) {
        if (dbg & 64) return false;  // hi_only: q_hi products only (the caller's error bound then carries |q_lo| |e|, api.hip)
        if constexpr (MODE == 0) return has_q && __builtin_amdgcn_readfirstlane(__float_as_int(qmeta[2 * (has_q ? wv : 0) + 1])) != 0;
        else return group_of(t) < n_q && __builtin_amdgcn_readfirstlane(__float_as_int(rs.q_anylo[group_of(t) < n_q ? group_of(t) : 0])) != 0;
    };
    bool any_lo = any_lo_of(0);

    // ---- accumulators: S^T[query vector 16 qb + 4 g + u][corpus row 16 a + j], lane = 16 g + j ------------------------
    f32x4 acc[NQB][MG_NBLK];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int a = 0; a < MG_NBLK; ++a) acc[qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float carry[NQB * 4];  // scanned maxima of the previous block (lane 15 of each DPP row = the chunk still open)
#pragma unroll
    for (int r = 0; r < NQB * 4; ++r) carry[r] = -INFINITY;
    uint32_t prev_last_end = 1;  // is the row before the tile's first row the last row of its chunk?

    const uint32_t sw = (uint32_t)((fj >> 1) & 7);
    const uint32_t a_hi = (uint32_t)(fj * 128) + ((kq ^ sw) << 4), a_lo = (uint32_t)(fj * 128) + (((4 + kq) ^ sw) << 4);
    float* const outq = out + (int64_t)((MODE == 0 && has_q) ? wv : 0) * out_stride;

    // ---- one K slab.  Fragments of 16-row blocks are read a PAIR ahead; the first pair of a slab is read at the end of the
    // previous slab (the barrier at the top of slab g certifies slab g + 1 as landed), so the MFMAs start right after the
    // barrier instead of after an LDS round trip that both waves of a SIMD would sit out together. -------------------------
    int c_tile = 0, c_s = 0, c_slot = 0;
    h16x8 eh[2][2], el[2][2];  // [pair parity][block of the pair]
    auto read_pair = [&](int slot, int p, h16x8 (&h)[2], h16x8 (&l)[2]) __attribute__((always_inline)) {
        const char* const base = smem + slot * SLAB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (HALF) {
                h[i] = *reinterpret_cast<const h16x8*>(base + (2 * p + i) * BLKB + lane16);
            } else {
                h[i] = *reinterpret_cast<const h16x8*>(base + (2 * p + i) * BLKB + a_hi);
                l[i] = *reinterpret_cast<const h16x8*>(base + (2 * p + i) * BLKB + a_lo);
            }
        }
    };
    auto slab = [&](f32x4 (&q)[4], f32x4 (&qn)[4], int nb, int g_trace) __attribute__((always_inline)) {
        Feed f{};
        if (feeder) f = next_feed();
        const int next_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
        if (has_q || MODE != 0) {
            // The next slab's query fragments: ONE load after each of the first four pairs.  (All four at the top of the slab
            // stalled both waves of every SIMD for ~500 cycles while the matrix pipe idled: a VMEM instruction costs its wave
            // 60-185 cycles of issue, profiles/r02_e_trace.txt.)
            const char* const qp = qbase + (int64_t)q_s * 4096;
            q_advance();
            h16x8 qh[2], ql[2];
            __builtin_memcpy(&qh[0], &q[0], 16);
            __builtin_memcpy(&ql[0], &q[1], 16);
            __builtin_memcpy(&qh[1], &q[2], 16);
            __builtin_memcpy(&ql[1], &q[3], 16);
#pragma unroll
            for (int p = 0; p < MG_NBLK / 2; ++p) {
                if (!(dbg & 8)) {
                    if (p + 1 < MG_NBLK / 2) read_pair(c_slot, p + 1, eh[(p + 1) & 1], el[(p + 1) & 1]);
                    else read_pair(next_slot, 0, eh[0], el[0]);
                }
                if (2 * p < nb && has_q && !(dbg & 2)) {  // wave-uniform: blocks past the workgroup's range are not multiplied
                    const h16x8(&h)[2] = eh[p & 1];
                    const h16x8(&l)[2] = el[p & 1];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            acc[qb][2 * p + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[qb], h[i], acc[qb][2 * p + i], 0, 0, 0);
                    if constexpr (!HALF) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int qb = 0; qb < NQB; ++qb)
                                acc[qb][2 * p + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[qb], l[i], acc[qb][2 * p + i], 0, 0, 0);
                    }
                    if (any_lo) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int qb = 0; qb < NQB; ++qb)
                                acc[qb][2 * p + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[qb], h[i], acc[qb][2 * p + i], 0, 0, 0);
                    }
                }
                if (p < 4) mg_load_frag(qn[p], lane16, qp + p * 1024);
                if (feeder && p >= 4 && !(dbg & (4 | 32))) {  // two DMAs after each of the last four pairs (p is a constant after unrolling)
                    [&]<int... P>(std::integer_sequence<int, P...>) { ((P / 2 == p - 4 ? dma_piece(f, std::integral_constant<int, P>{}) : (void)0), ...); }
                    (std::make_integer_sequence<int, 8>{});
                }
                stamp(g_trace, 3 + p);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (feeder && (dbg & 32)) dma_all(f);
        } else if (feeder) {
            dma_all(f);
        }
        c_slot = next_slot;
    };

    // HO: q = the hi fragments of this slab (qb0, qb1); qn receives those of the slab three ahead -- one load after each of the
    // first two pairs, the feeders' four DMAs after the last four.  Every wave loads and reads (a wave without a query multiplies
    // nothing): the vmcnt bookkeeping of the main loop is the same for all of them.
    [[maybe_unused]] auto slab_ho = [&](f32x4 (&q)[2], f32x4 (&qn)[2], int nb) __attribute__((always_inline)) {
        Feed f{};
        if (feeder) f = next_feed();
        const int next_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
        const char* const qp = qbase + (int64_t)q_s * 4096;  // slab q_s: [qb0 hi | qb0 lo | qb1 hi | qb1 lo] x 1 KiB
        q_advance();
        h16x8 qh[2];
        __builtin_memcpy(&qh[0], &q[0], 16);
        __builtin_memcpy(&qh[1], &q[1], 16);
#pragma unroll
        for (int p = 0; p < MG_NBLK / 2; ++p) {
            if (p + 1 < MG_NBLK / 2) read_pair(c_slot, p + 1, eh[(p + 1) & 1], el[(p + 1) & 1]);
            else read_pair(next_slot, 0, eh[0], el[0]);
            if (2 * p < nb && has_q) {  // wave-uniform
                const h16x8(&h)[2] = eh[p & 1];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
                        acc[qb][2 * p + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[qb], h[i], acc[qb][2 * p + i], 0, 0, 0);
            }
            if (p < 2) mg_load_frag(qn[p], lane16, qp + p * 2048);
            if (feeder && p >= 4) {
                [&]<int... P>(std::integer_sequence<int, P...>) { ((P / 2 == p - 4 ? dma_piece(f, std::integral_constant<int, P>{}) : (void)0), ...); }
                (std::make_integer_sequence<int, 8>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        c_slot = next_slot;
    };

    // ---- tile epilogue: per-chunk maxima along the DPP rows, sum over the query vectors, store ---------------------------
    auto epilogue_rows = [&](int t) __attribute__((always_inline)) {  // MODE 1: metric + store of the tile's 32 x 256 scores
        const int32_t row0 = tile_row0(t);
        const int64_t col0 = rs.compact ? (int64_t)(row0 / (stride * MG_TM)) * MG_TM - row0 : 0;  // column of S = col0 + row
        if (!has_q) return;
        const int g = lane >> 4;
        const int32_t q0 = group_of(t) * 32 + 4 * g;
        const int mode = rs.metric;
        const bool norms = mode == SCAN_COSINE || mode == SCAN_L2;
        float us[NQB * 4], qss[NQB * 4], qn[NQB * 4];
#pragma unroll
        for (int r = 0; r < NQB * 4; ++r) {
            const int32_t q = q0 + 16 * (r >> 2) + (r & 3);
            const int32_t qc = q < rs.B ? q : rs.B - 1;
            us[r] = rs.q_unscale[qc] * inv_e_scale;  // powers of two: exact
            qss[r] = norms ? rs.q_sumsq[qc] : 0.f;
            qn[r] = sqrtf(qss[r]);
        }
#pragma unroll
        for (int a = 0; a < MG_NBLK; ++a) {
            const int32_t row = row0 + 16 * a + fj;
            const float rn = norms ? rs.row_norm[row < (int32_t)n_rows ? row : (int32_t)n_rows - 1] : 0.f;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * qb + u;
                    const int32_t q = q0 + 16 * qb + u;
                    const float d = acc[qb][a][u] * us[r];
                    float v = d;  // SCAN_RAW_DOT
                    // the formulas (and operation order) of scan.hip:transform_kernel / score_gemm.hip: same bits
                    if (mode == SCAN_COSINE) v = 1.0f - (1.0f - d / (rn * qn[r]));
                    else if (mode == SCAN_DOT) v = 1.0f + d;
                    else if (mode == SCAN_L2) v = 1.0f - sqrtf(fmaxf(rn + qss[r] - 2.0f * d, 0.f));
                    // (compact sample layout: every column of S belongs to a sampled tile -- rows past the corpus in the last one
                    // are written as -inf here, so the caller needs no fill pass over the sample matrix)
                    if (q < rs.B && (row < (int32_t)n_rows || rs.compact)) rs.S[(int64_t)q * rs.ld + col0 + row] = row < (int32_t)n_rows ? v : -INFINITY;
                }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) acc[qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // MODE 2: no score matrix.  A score that reaches its query's threshold (the k-th best of a row sample: a lower bound of
    // the k-th best overall, so every row of the exact top-k passes) is kept as a record in a wave-private LDS list and
    // appended to the query's candidate list when the tile is done; everything else is dropped after one compare.  The cosine
    // test uses reciprocals (2 VALU instead of an IEEE divide) against a threshold lowered by 1e-5; the exact similarity --
    // the formulas of MODE 1, same bits -- is computed for the few records only.
    [[maybe_unused]] int ncand = 0;  // wave-uniform
    [[maybe_unused]] float* const rec_d = reinterpret_cast<float*>(smem + NSLOT * SLAB + (TRACE ? 16 * 8 * 16 * 8 : 0) + wv * 4096);
    [[maybe_unused]] uint32_t* const rec_m = reinterpret_cast<uint32_t*>(rec_d + 512);
    constexpr int REC_CAP = 512;  // records a wave keeps per tile (expected: a few dozen); more -> the guarded dense fallback
    auto flush_candidates = [&](int t) __attribute__((always_inline)) {
        const int32_t row0 = tile_row0(t);
        const int mode = rs.metric;
        for (int i = lane; i < ncand; i += 64) {
            const uint32_t m = rec_m[i];
            const int32_t q = group_of(t) * 32 + (int32_t)(m >> 8), row = row0 + (int32_t)(m & 255u);
            const float d = rec_d[i] * (rs.q_unscale[q] * inv_e_scale);  // the dot product, as MODE 1 forms it
            float v = d;
            if (mode == SCAN_COSINE) v = 1.0f - (1.0f - d / (rs.row_norm[row] * sqrtf(rs.q_sumsq[q])));
            else if (mode == SCAN_DOT) v = 1.0f + d;
            const uint32_t slot = atomicAdd(rs.cand_cnt + q, 1u);
            if (slot < (uint32_t)rs.cap) {
                rs.cand_scores[(int64_t)q * rs.cap + slot] = v;
                rs.cand_ids[(int64_t)q * rs.cap + slot] = row;
            } else {
                *rs.overflow = 1u;
            }
        }
        ncand = 0;
    };
    auto epilogue_cand = [&](int t) __attribute__((always_inline)) {
        if (!has_q) return;
        const int32_t row0 = tile_row0(t);
        // Opaque copies of the lane coordinates: with the plain values every record word below (128 of them) is loop-invariant,
        // the compiler hoists all of them out of the K loop and spills them -- and a scratch reload anywhere in the loop makes
        // its waitcnt pass put s_waitcnt vmcnt(0) into the slab code, which drains the look-ahead DMAs three times per two slabs.
        int g = lane >> 4, fj = lane & 15;
        asm volatile("" : "+v"(g), "+v"(fj));
        const int32_t q0 = group_of(t) * 32 + 4 * g;
        const bool cosine = rs.metric == SCAN_COSINE;
        // One threshold per accumulator register, on the RAW accumulator (times 1 / |e| for cosine): a lower bound of what
        // the exact formula needs -- 1e-5 of slack covers the reciprocals and the two roundings of 1 - (1 - c) -- so the test
        // costs one multiply and one compare, and a record that passes is re-evaluated exactly at the flush.
        float thr[NQB * 4];
        bool bad = false;
#pragma unroll
        for (int r = 0; r < NQB * 4; ++r) {
            const int32_t q = q0 + 16 * (r >> 2) + (r & 3);
            const int32_t qc = q < rs.B ? q : rs.B - 1;
            const float us = rs.q_unscale[qc] * inv_e_scale;  // > 0
            const float tq = rs.tau[(int64_t)qc * rs.tau_stride];
            bad |= !(tq > -INFINITY);  // NaN or -inf: fewer than k usable sample scores
            const float slack = 1e-5f * fmaxf(1.0f, fabsf(tq));
            const float t_dot = cosine ? (tq - slack) * sqrtf(rs.q_sumsq[qc]) : (tq - 1.0f) - slack;  // bound on d (cosine: on d / |e|)
            thr[r] = q < rs.B ? (t_dot - fabsf(t_dot) * 1e-6f) / us : INFINITY;
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *rs.overflow = 1u;
#pragma unroll
        for (int a = 0; a < MG_NBLK; ++a) {
            const int32_t row = row0 + 16 * a + fj;
            const bool row_ok = row < (int32_t)n_rows;
            const float rscale = cosine ? __builtin_amdgcn_rcpf(rs.row_norm[row_ok ? row : (int32_t)n_rows - 1]) : 1.0f;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = 4 * qb + u;
                    const bool pass = row_ok && acc[qb][a][u] * rscale >= thr[r];
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                    if (mask != 0ull) {  // wave-uniform, rare
                        const int pos = ncand + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                        if (pass && pos < REC_CAP) {
                            rec_d[pos] = acc[qb][a][u];
                            rec_m[pos] = ((uint32_t)(16 * qb + 4 * g + u) << 8) | (uint32_t)(16 * a + fj);
                        }
                        ncand += __builtin_popcountll(mask);
                    }
                }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) acc[qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);  // one block at a time: hoisting the 16 norm loads costs registers this kernel does not have
        }
        if (ncand > REC_CAP) {  // (wave-uniform) more than the wave can hold: let the dense path decide
            if (lane == 0) *rs.overflow = 1u;
            ncand = REC_CAP;
        }
        if (ncand > 0) flush_candidates(t);
    };
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int32_t row0 = org + t * MG_TM;
        // "last row of its chunk" bits of the tile's 256 rows: 9 words from row0 / 32, shifted by 16 when row0 is odd in blocks
        uint32_t m[9];
        const uint32_t* const eb = ends_bits + (row0 >> 5);
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = eb[i];
        const bool odd = (row0 & 16) != 0;
        uint32_t mm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mm[i] = odd ? (m[i] >> 16) | (m[i + 1] << 16) : m[i];
        if (has_q && !(dbg & 1)) {
#pragma unroll
            for (int a = 0; a < MG_NBLK; ++a) {
                const uint32_t E = (mm[a >> 1] >> (16 * (a & 1))) & 0xffffu;
                const int32_t base = row0 + 16 * a;
                const int32_t ord0 = row_to_chunk[base < (int32_t)n_rows ? base : (int32_t)n_rows];
                // lanes whose shifted neighbour belongs to the same chunk: no chunk end in rows [j - d, j - 1]
                const uint32_t O1 = E << 1, O2 = O1 | (O1 << 1), O4 = O2 | (O2 << 2), O8 = O4 | (O4 << 4);
                const uint64_t rep = 0x0001000100010001ull;
                const uint64_t F1 = (uint64_t)(~O1 & 0xfffeu) * rep, F2 = (uint64_t)(~O2 & 0xfffcu) * rep;
                const uint64_t F4 = (uint64_t)(~O4 & 0xfff0u) * rep, F8 = (uint64_t)(~O8 & 0xff00u) * rep;
                const uint64_t C0 = prev_last_end ? 0ull : rep;  // row 0 continues the chunk open at the end of the previous block
                float x[NQB * 4];
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[4 * qb + u] = acc[qb][a][u];
                float tmp[NQB * 4];
#define MG_STEP(SRC, CTRLSTR, MASK)                                                        \
    _Pragma("unroll") for (int r = 0; r < NQB * 4; ++r) MG_MAX_DPP(tmp[r], SRC, x[r], CTRLSTR); \
    _Pragma("unroll") for (int r = 0; r < NQB * 4; ++r) MG_SELECT(x[r], tmp[r], MASK);
                // (skipping a step wave-uniformly when no lane takes it measured SLOWER: 17-20 k cycles per tile against 15.5-18 k)
                MG_STEP(carry[r], "row_ror:1", C0)  // lane 0 <- lane 15 of the previous block's scan
                MG_STEP(x[r], "row_shr:1", F1)
                MG_STEP(x[r], "row_shr:2", F2)
                MG_STEP(x[r], "row_shr:4", F4)
                MG_STEP(x[r], "row_shr:8", F8)
#undef MG_STEP
#pragma unroll
                for (int r = 0; r < NQB * 4; ++r) carry[r] = x[r];
                // rows of this block inside the workgroup's range
                int32_t lo = r_lo - base, hi = r_hi - base;
                lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                const uint32_t EM = E & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                if (EM != 0u) {  // wave-uniform: some chunk of this workgroup ends in this block
                    float tsum;
                    if constexpr (NQB == 2) tsum = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                    else tsum = (x[0] + x[1]) + (x[2] + x[3]);
                    tsum += __shfl_xor(tsum, 16);
                    tsum += __shfl_xor(tsum, 32);
                    const uint32_t below = E & ((1u << fj) - 1u);
                    if (lane < 16 && ((EM >> lane) & 1u)) outq[ord0 + __builtin_popcount(below)] = tsum * unscale;
                }
                prev_last_end = (E >> 15) & 1u;
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) acc[qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
            prev_last_end = (mm[7] >> 31) & 1u;
        }
    };

    if constexpr (HO) {
        // ---- HO main loop: four slabs per iteration (four static sets of two query fragments) --------------------------------------
        // Issue order per slab g: Q(g + 3) (2 loads), then -- feeders -- the 4 DMAs of slab g + 5.  At the top of slab g a wave needs
        // Q(g) and, for everybody's sake, its DMAs of slab g + 1: everything up to Q(g) has retired when at most DMA(g + 2), Q(g + 1),
        // DMA(g + 3), Q(g + 2), DMA(g + 4) = 16 operations are outstanding (the other waves: Q(g + 1), Q(g + 2) = 4).
        auto tile_nb = [&](int t) {
            const int32_t left = r_hi - tile_row0(t);
            const int nb_ = (left + 15) >> 4;
            return nb_ < MG_NBLK ? nb_ : MG_NBLK;
        };
        int nb = tile_nb(0);
        auto advance = [&]() __attribute__((always_inline)) {
            if (++c_s == nslab) {
                if constexpr (MODE == 0) epilogue(c_tile);
                else if constexpr (MODE == 1) epilogue_rows(c_tile);
                else epilogue_cand(c_tile);
                c_s = 0;
                ++c_tile;
                nb = tile_nb(c_tile);
                if constexpr (MODE != 0) has_q = group_of(c_tile) < n_q;  // the next tile's group of queries
            }
        };
        f32x4 qs0[2], qs1[2], qs2[2], qs3[2];
        auto load_q2 = [&](f32x4 (&q)[2]) __attribute__((always_inline)) {
            const char* p = qbase + (int64_t)q_s * 4096;
            mg_load_frag(q[0], lane16, p);
            mg_load_frag(q[1], lane16, p + 2048);
            q_advance();
        };
        {   // prologue: DMA(0), DMA(1), Q(0), DMA(2), Q(1), DMA(3), Q(2), DMA(4): the steady-state order
            if (feeder) { const Feed f0 = next_feed(); dma_all(f0); const Feed f1 = next_feed(); dma_all(f1); }
            load_q2(qs0);
            if (feeder) { const Feed f2 = next_feed(); dma_all(f2); }
            load_q2(qs1);
            if (feeder) { const Feed f3 = next_feed(); dma_all(f3); }
            load_q2(qs2);
            if (feeder) { const Feed f4 = next_feed(); dma_all(f4); asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); }  // DMA(0) has landed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            read_pair(0, 0, eh[0], el[0]);
        }
        auto step = [&](f32x4 (&q)[2], f32x4 (&qn)[2]) __attribute__((always_inline)) {
            if (feeder) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            asm volatile("" : "+v"(q[0]), "+v"(q[1]));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            slab_ho(q, qn, nb);
            advance();
        };
        for (int g = 0; g < total; g += 4) {
            step(qs0, qs3);
            if (g + 1 < total) step(qs1, qs0);
            if (g + 2 < total) step(qs2, qs1);
            if (g + 3 < total) step(qs3, qs2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
        return;
    }
    // ---- main loop: two slabs per iteration (two static sets of query fragment registers) ---------------------------------
    // VMEM retires in order.  Per slab g a feeder issues Q(g + 1) (first half of the slab), then its 8 DMAs of slab g + 3; at the top of slab g it
    // needs DMA(g + 1) and Q(g), i.e. at most the 8 DMAs of slab g + 2 outstanding.  The other waves only have Q in flight.
    f32x4 qa[4], qb_[4];
    {   // prologue: DMA(0), DMA(1), Q(0), DMA(2); slab 0 must have landed for everybody before its first pair is read
        if (feeder) { const Feed f0 = next_feed(); dma_all(f0); const Feed f1 = next_feed(); dma_all(f1); }
        if (has_q || MODE != 0) load_q(qa);
        if (feeder) { const Feed f2 = next_feed(); dma_all(f2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HALF ? 12 : 20) : "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (has_q || MODE != 0) read_pair(0, 0, eh[0], el[0]);
    }
    auto tile_nb = [&](int t) {
        const int32_t left = r_hi - tile_row0(t);
        const int nb = (left + 15) >> 4;
        return nb < MG_NBLK ? nb : MG_NBLK;
    };
    int nb = tile_nb(0);
    auto advance = [&]() __attribute__((always_inline)) {
        if (++c_s == nslab) {
            if constexpr (MODE == 0) epilogue(c_tile);
            else if constexpr (MODE == 1) epilogue_rows(c_tile);
            else epilogue_cand(c_tile);
            c_s = 0;
            ++c_tile;
            nb = tile_nb(c_tile);
            if constexpr (MODE != 0) {  // the next tile's group of queries
                has_q = group_of(c_tile) < n_q;
                any_lo = any_lo_of(c_tile);
            }
        }
    };
    auto wait_top = [&]() __attribute__((always_inline)) {
        if (feeder) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HALF ? 4 : 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto run_slab = [&](f32x4 (&q)[4], f32x4 (&qn)[4], int g_trace) __attribute__((always_inline)) { slab(q, qn, nb, g_trace); };
    for (int g = 0; g < total; g += 2) {
        stamp(g, 0);
        wait_top();
        mg_pin(qa[0], qa[1], qa[2], qa[3]);
        stamp(g, 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(g, 2);
        run_slab(qa, qb_, g);
        advance();
        stamp(g, 11);
        if (g + 1 < total) {
            stamp(g + 1, 0);
            wait_top();
            mg_pin(qb_[0], qb_[1], qb_[2], qb_[3]);
            stamp(g + 1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            stamp(g + 1, 2);
            run_slab(qb_, qa, g + 1);
            advance();
            stamp(g + 1, 11);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
    if constexpr (TRACE) {
        __syncthreads();
        if (blockIdx.x == 7)
            for (int i = threadIdx.x; i < 16 * 8 * 16; i += blockDim.x) trace[i] = reinterpret_cast<unsigned long long*>(smem + NSLOT * SLAB)[i];
    }
}

// n_q (1..8) queries `first .. first + n_q - 1` of a launch_query_planes buffer over `n_queries`, each nq (<= 32) vectors:
// out[q * out_stride + chunk].  Needs an index without empty chunks (the chunk of an end row is found by counting ends).
int launch_maxsim_gemm(const void* planes, int64_t n_rows, int32_t dim, const void* qbuf, int32_t n_queries, int32_t first,
                       int32_t n_q, int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, const uint32_t* ends_bits,
                       float* out, int64_t out_stride, int n_cu, hipStream_t s, float split_scale, bool half, const uint32_t* run_if,
                       bool hi_only, bool all_passes) {
    // all_passes: n_q may exceed 8 -- ceil(n_q / 8) passes in ONE launch (gridDim.y), pass y writing out + 8 y out_stride
    if (nq < 1 || nq > 32 || n_q < 1 || (n_q > MG_WAVES && !all_passes) || n_rows < 1 || first < 0 || first + n_q > n_queries) return RL_ERR_UNSUPPORTED;
    if (dim % 32 || dim < 32 || !(split_scale > 0.f) || !planes || !ends_bits) return RL_ERR_UNSUPPORTED;
    const int32_t nslab = dim / 32;
    const char* qfrag = static_cast<const char*>(qbuf) + (size_t)first * nslab * 4096;
    const float* qmeta = reinterpret_cast<const float*>(static_cast<const char*>(qbuf) + (size_t)n_queries * dim * 128) + 2 * (size_t)first;
    const int64_t tiles = (n_rows + MG_TM - 1) / MG_TM;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles)), (unsigned)((n_q + MG_WAVES - 1) / MG_WAVES)), blk(512);
    static const int dbg_env = exp_env("RAGLITE_GEMM_DBG") ? std::atoi(exp_env("RAGLITE_GEMM_DBG")) : 0;  // timing experiments only
    const int dbg = (dbg_env & ~64) | (hi_only ? 64 : 0);
#ifdef RAGLITE_EXPERIMENTS  // the slab-timeline build of the kernel exists in experiment builds only
    static unsigned long long* trace = [] {
        unsigned long long* p = nullptr;
        if (exp_env("RAGLITE_GEMM_TRACE")) { (void)hipMalloc(&p, 16 * 8 * 16 * 8); (void)hipMemset(p, 0, 16 * 8 * 16 * 8); }
        return p;
    }();
    if (trace && nq > 16 && !half) {  // diagnostic build: dump the 10th launch's slab timeline to stderr
        static int calls = 0;
        hipLaunchKernelGGL((maxsim_gemm_kernel<2, true>), grid, blk, 0, s, static_cast<const char*>(planes), n_rows, nslab, qfrag, qmeta, n_q,
                           row_to_chunk, chunk_offsets, ends_bits, out, out_stride, 1.0f / split_scale, dbg, trace, RowScoreArgs{});
        if (++calls == 10) {
            static unsigned long long h[16 * 8 * 16];
            (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "GEMMTRACE columns: before-vmcnt after-vmcnt after-barrier after-pair0..7 after-advance(epilogue)\n");
            for (int g = 0; g < 16; ++g)
                for (int wv = 0; wv < 8; ++wv) {
                    fprintf(stderr, "GEMMTRACE slab %d wave %d:", g + 128, wv);
                    for (int k = 0; k < 12; ++k) fprintf(stderr, " %7lld", (long long)(h[(g * 8 + wv) * 16 + k] - h[0]));
                    fprintf(stderr, "\n");
                }
        }
        return RL_OK;
    }
#endif
    RowScoreArgs rs0{};
    rs0.run_if = run_if;
#define RL_MG_LAUNCH(NQB_, HALF_)                                                                                                     \
    hipLaunchKernelGGL((maxsim_gemm_kernel<NQB_, false, 0, HALF_>), grid, blk, 0, s, static_cast<const char*>(planes), n_rows, nslab, qfrag, \
                       qmeta, n_q, row_to_chunk, chunk_offsets, ends_bits, out, out_stride, 1.0f / split_scale, dbg, nullptr, rs0)
    if (half && hi_only) {  // one product over a one-plane image: the deep-stream build (HO)
#define RL_MG_LAUNCH_HO(NQB_)                                                                                                          \
    hipLaunchKernelGGL((maxsim_gemm_kernel<NQB_, false, 0, true, true>), grid, blk, 0, s, static_cast<const char*>(planes), n_rows, nslab, qfrag, \
                       qmeta, n_q, row_to_chunk, chunk_offsets, ends_bits, out, out_stride, 1.0f / split_scale, dbg, nullptr, rs0)
        if (nq <= 16) RL_MG_LAUNCH_HO(1); else RL_MG_LAUNCH_HO(2);
#undef RL_MG_LAUNCH_HO
    }
    else if (half) { if (nq <= 16) RL_MG_LAUNCH(1, true); else RL_MG_LAUNCH(2, true); }
    else      { if (nq <= 16) RL_MG_LAUNCH(1, false); else RL_MG_LAUNCH(2, false); }
#undef RL_MG_LAUNCH
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- MODE 1: many independent queries (a6 batched, BASELINE cfg 5) -------------------------------------------------------
// Query image: groups of 32 queries in the fragment layout of query_planes_kernel, but every query scaled by a power of two
// of its OWN (a batch may span many orders of magnitude): unscale[q] = 2^(ex_q - 14), anylo[group].
__global__ __launch_bounds__(1024) void query_rows_planes_kernel(const float* __restrict__ Q, int B, int dim, uint4* __restrict__ frag,
                                                                  float* __restrict__ unscale, float* __restrict__ anylo, uint32_t* __restrict__ zero_word) {
    __shared__ float scale_sh[32];
    __shared__ int any_lo_sh;
    const int grp = blockIdx.x, nslab = dim >> 5;
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;  // (the search's overflow flag: no memset launch in front of the batch)
    const int v = threadIdx.x >> 5, sub = threadIdx.x & 31;  // 32 threads per query for the magnitude (a group of 32 queries per block: B / 32 blocks
                                                             // of 256 threads were 24 us of latency per 1000-query batch)
    const int q = grp * 32 + v;
    float mx = 0.f;
    if (q < B)
        for (int c = 4 * sub; c < dim; c += 128) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(Q + (int64_t)q * dim + c);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))));
        }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    mx = fmaxf(mx, __shfl_xor(mx, 8));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);
    ex = ex > -100 ? ex : -100;
    if (sub == 0) {
        scale_sh[v] = ldexpf(1.f, 14 - ex);
        if (q < B) unscale[q] = ldexpf(1.f, ex - 14);
    }
    if (threadIdx.x == 0) any_lo_sh = 0;
    __syncthreads();
    bool any_lo = false;
    uint4* const out = frag + (int64_t)grp * nslab * 4 * 64;
    for (int t = threadIdx.x; t < nslab * 2 * 64; t += 1024) {
        const int lane = t & 63, qb = (t >> 6) & 1, s = t >> 7;
        const int vi = 16 * qb + (lane & 15), kq = lane >> 4;
        const int qi = grp * 32 + vi;
        h16x8 hi8, lo8;
        f32x4 v0 = (f32x4){0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (qi < B) {
            const float* p = Q + (int64_t)qi * dim + 32 * s + 8 * kq;
            v0 = *reinterpret_cast<const f32x4*>(p);
            v1 = *reinterpret_cast<const f32x4*>(p + 4);
        }
        const float sc = scale_sh[vi];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = (u < 4 ? v0[u] : v1[u - 4]) * sc;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)hi);
            hi8[u] = hi;
            lo8[u] = lo;
            any_lo |= lo != (_Float16)0.0f;
        }
        uint4 a, b;
        __builtin_memcpy(&a, &hi8, 16);
        __builtin_memcpy(&b, &lo8, 16);
        out[(s * 4 + 2 * qb + 0) * 64 + lane] = a;
        out[(s * 4 + 2 * qb + 1) * 64 + lane] = b;
    }
    if (any_lo) any_lo_sh = 1;  // benign race: every writer stores 1
    __syncthreads();
    if (threadIdx.x == 0) anylo[grp] = any_lo_sh ? 1.f : 0.f;
}

// floats of scratch for `nb` queries: fragments (dim * 32 per group of 32), unscale[nb], anylo[groups], q_sumsq[nb]
size_t score_planes_scratch_floats(int32_t nb, int32_t dim) {
    const size_t groups = ((size_t)nb + 31) / 32;
    return groups * 32 * (size_t)dim + 2 * (size_t)nb + groups + 64;
}

int launch_query_sumsq(const float* Q, int32_t nb, int32_t dim, float* out, hipStream_t s);  // score_gemm.hip

// Query side of the row-score modes: fragments, per-query unscale, per-group lo flags, |q|^2 -> `scratch`
// (score_planes_scratch_floats floats, 16-B aligned).
int launch_score_planes_queries(const float* Q, int32_t nb, int32_t dim, float* scratch, int mode, hipStream_t s, uint32_t* zero_word) {
    if (nb < 1 || dim % 32 || dim < 32) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Q) & 15) || (reinterpret_cast<uintptr_t>(scratch) & 15)) return RL_ERR_UNSUPPORTED;
    const int32_t groups = (nb + 31) / 32;
    float* frag = scratch;
    float* unscale = frag + (size_t)groups * 32 * dim;
    float* anylo = unscale + nb;
    float* qss = anylo + groups;
    hipLaunchKernelGGL(query_rows_planes_kernel, dim3((unsigned)groups), dim3(1024), 0, s, Q, (int)nb, (int)dim, reinterpret_cast<uint4*>(frag), unscale,
                       anylo, zero_word);
    RL_HIP(hipGetLastError());
    if (mode == SCAN_COSINE || mode == SCAN_L2) RL_TRY(launch_query_sumsq(Q, nb, dim, qss, s));
    return RL_OK;
}

// One pass of the row-score GEMM over the pre-split corpus image with the query side prepared by
// launch_score_planes_queries.  Dense (cand == nullptr): scores[q * ld + row] (or + sampled-tile column with
// tile_stride > 1).  Fused top-k (cand != nullptr): candidate lists, see RowScoreArgs.
int launch_score_planes_pass(const void* planes, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, float* scores, int64_t ld,
                             const float* row_norm, const float* row_sumsq, int mode, int32_t tile_stride, const uint32_t* run_if,
                             const CandArgs* cand, int n_cu, hipStream_t s, float split_scale, bool half, bool hi_only) {
    if (nb < 1 || n_rows < 1 || dim % 32 || dim < 32 || !(split_scale > 0.f) || !planes || tile_stride < 1) return RL_ERR_UNSUPPORTED;
    if ((mode == SCAN_COSINE && !row_norm) || (mode == SCAN_L2 && !row_sumsq)) return RL_ERR_INVALID;
    const int32_t nslab = dim / 32, groups = (nb + 31) / 32;
    float* frag = scratch;
    float* unscale = frag + (size_t)groups * 32 * dim;
    float* anylo = unscale + nb;
    float* qss = anylo + groups;
    RowScoreArgs rs{};
    rs.S = scores; rs.ld = ld; rs.row_norm = mode == SCAN_COSINE ? row_norm : row_sumsq; rs.q_sumsq = qss; rs.q_unscale = unscale;
    rs.q_anylo = anylo; rs.B = nb; rs.QT = (groups + MG_WAVES - 1) / MG_WAVES; rs.metric = mode; rs.tile_stride = tile_stride;
    rs.compact = tile_stride > 1 ? 1 : 0; rs.run_if = run_if;
    static const int q_inner_env = exp_env("RAGLITE_GEMM_Q_INNER") ? 1 : 0;  // A/B: query tile fastest (the first version)
    rs.q_outer = q_inner_env ? 0 : 1;
    if (cand) { rs.tau = cand->tau; rs.tau_stride = cand->tau_stride; rs.cand_scores = cand->scores; rs.cand_ids = cand->ids; rs.cand_cnt = cand->cnt;
                rs.overflow = cand->overflow; rs.cap = cand->cap; }
    const int64_t tiles = ((n_rows + MG_TM - 1) / MG_TM + tile_stride - 1) / tile_stride;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles * rs.QT))), blk(512);
#define RL_MG_LAUNCH(MODE_, HALF_)                                                                                                      \
    hipLaunchKernelGGL((maxsim_gemm_kernel<2, false, MODE_, HALF_>), grid, blk, 0, s, static_cast<const char*>(planes), n_rows, nslab,          \
                       reinterpret_cast<const char*>(frag), nullptr, groups, nullptr, nullptr, nullptr, nullptr, (int64_t)0, 1.0f / split_scale, \
                       hi_only ? 64 : 0, nullptr, rs)
    if (half && hi_only) {  // one product over a one-plane image: the deep-stream build (HO)
#define RL_MG_LAUNCH_HO(MODE_)                                                                                                         \
    hipLaunchKernelGGL((maxsim_gemm_kernel<2, false, MODE_, true, true>), grid, blk, 0, s, static_cast<const char*>(planes), n_rows, nslab,     \
                       reinterpret_cast<const char*>(frag), nullptr, groups, nullptr, nullptr, nullptr, nullptr, (int64_t)0, 1.0f / split_scale, \
                       64, nullptr, rs)
        if (cand) RL_MG_LAUNCH_HO(2); else RL_MG_LAUNCH_HO(1);
#undef RL_MG_LAUNCH_HO
    }
    else if (half) { if (cand) RL_MG_LAUNCH(2, true); else RL_MG_LAUNCH(1, true); }
    else      { if (cand) RL_MG_LAUNCH(2, false); else RL_MG_LAUNCH(1, false); }
#undef RL_MG_LAUNCH
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Similarity (metric `mode`, scan.hip conventions) of nb queries against every row over the pre-split corpus image:
// scores[q * ld + row].  Same results as launch_score_gemm in split arithmetic up to the summation order over K.
int launch_score_planes(const void* planes, int64_t n_rows, int32_t dim, const float* Q, int32_t nb, float* scores, int64_t ld,
                        const float* row_norm, const float* row_sumsq, float* scratch, int mode, int n_cu, hipStream_t s, float split_scale,
                        bool half) {
    if (!(split_scale > 0.f) || !planes) return RL_ERR_UNSUPPORTED;
    RL_TRY(launch_score_planes_queries(Q, nb, dim, scratch, mode, s));
    return launch_score_planes_pass(planes, n_rows, dim, nb, scratch, scores, ld, row_norm, row_sumsq, mode, 1, nullptr, nullptr, n_cu, s,
                                    split_scale, half);
}

}  // namespace rl
