// a9: MaxSim for shapes outside the dim == 1024 streaming fast path.
//
//   maxsim_generic_kernel : any dim <= 4096, any nq <= 1024; one wave per (query, chunk) item; VALU dots
//                           with a wave butterfly per (row, query vector).  Correctness backstop.
//   maxsim_pairs_kernel, maxsim_pairs_packed_kernel, maxsim_pairs_wide_kernel : exact fp32-MFMA MaxSim of arbitrary (query, chunk) pairs --
//                           the re-scoring step of the bound-filtered batch and the rerank beyond dim 128 (dim % 16 up to 1024; % 128 up to 4096).
//   maxsim_cand_kernel    : the rerank shape (SURVEY.md cfg 3): dim == 128, nq <= 32, many independent
//                           queries x candidate lists per launch.  v_mfma_f32_16x16x4_f32 with the query's
//                           32 x 128 matrix resident in 64 VGPRs as B fragments; candidate rows are loaded
//                           straight into A-fragment layout (16 rows x 64 B per instruction = whole 64-B
//                           sectors, every byte fetched exactly once), no LDS: a chunk is only 16 rows x
//                           512 B per tile, and each wave handles 8 candidates of one query back to back.
//                           Algorithmic bytes/query = n_cand * rows * 512 B (8.39 MB at 256 x 64);
//                           algorithmic flops/query = 2 * nq * n_cand * rows * 128 (134 MFLOP).
//
// score[item] = sum_{i<nq} max_{j in chunk} Q[i].D[j]   (src/raglite/_search.py:143-149 generalised to
// several query vectors; plugged in behind the reranker call at src/raglite/_search.py:394-396).
#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int GEN_NQ_MAX = 1024;
constexpr int PAIRS_MAX_DIM = 4096;  // maxsim_pairs_*: <= 1024 with the query in LDS, beyond through wave-private windows

template <int NV, int VEC>
__global__ __launch_bounds__(256) void maxsim_generic_kernel(const float* __restrict__ D, int dim,
                                                              const float* __restrict__ Q, int nq,
                                                              const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ candidates,
                                                              int64_t n_items, float* __restrict__ out) {
    __shared__ float msh[4][GEN_NQ_MAX];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    float* m = msh[w];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + w;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t item = wave0; item < n_items; item += n_waves) {
        // one query per launch: item = chunk ordinal, or an index into this query's candidate list
        const int64_t chunk = candidates ? (int64_t)candidates[item] : item;
        // a negative ordinal = "no chunk" (padding of a search result, or sanitised by rl_maxsim_rerank): score -inf
        const int64_t b = chunk >= 0 ? offsets[chunk] : 0, e = chunk >= 0 ? offsets[chunk + 1] : 0;
        for (int i = lane; i < nq; i += 64) m[i] = -INFINITY;
        for (int64_t r = b; r < e; ++r) {
            float x[NV][VEC];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = (v * 64 + lane) * VEC;
#pragma unroll
                for (int j = 0; j < VEC; ++j) x[v][j] = 0.f;
                if (c < dim) {
                    if constexpr (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(D + r * (int64_t)dim + c);
                        x[v][0] = t.x; x[v][1] = t.y; x[v][2] = t.z; x[v][3] = t.w;
                    } else {
                        x[v][0] = D[r * (int64_t)dim + c];
                    }
                }
            }
            for (int i = 0; i < nq; ++i) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int c = (v * 64 + lane) * VEC;
                    if (c < dim) {
                        if constexpr (VEC == 4) {
                            const float4 t = *reinterpret_cast<const float4*>(Q + (int64_t)i * dim + c);
                            s = fmaf(x[v][0], t.x, s); s = fmaf(x[v][1], t.y, s);
                            s = fmaf(x[v][2], t.z, s); s = fmaf(x[v][3], t.w, s);
                        } else {
                            s = fmaf(x[v][0], Q[(int64_t)i * dim + c], s);
                        }
                    }
                }
                s = wave_sum(s);
                if (lane == 0) m[i] = fmaxf(m[i], s);
            }
        }
        // fixed-order sum over the query vectors: lane-strided partial sums, then a butterfly
        float t = 0.f;
        for (int i = lane; i < nq; i += 64) t += m[i];
        t = wave_sum(t);
        if (lane == 0) out[item] = (e > b) ? t : -INFINITY;
    }
}

template <int NV, int VEC>
static int generic_t(const float* D, int32_t dim, const float* Q, int32_t nq, const int64_t* offsets,
                     const int32_t* cand, int64_t n_items, float* out, hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_items + 3) / 4, 256 * 8));
    hipLaunchKernelGGL((maxsim_generic_kernel<NV, VEC>), dim3(blocks), dim3(256), 0, s, D, (int)dim, Q, (int)nq,
                       offsets, cand, n_items, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// maxsim_pairs_kernel: exact MaxSim of arbitrary (query, chunk) pairs at any dim % 16 == 0, dim <= 1024, nq <= 32 -- the
// rerank shape beyond dim 128 and the re-scoring step of the half-bytes MaxSim batch (api.hip: maxsim_batch_hi).
// One workgroup of eight waves serves ONE query (blockIdx.y; blockIdx.x splits its candidate list): the query's [32 x dim] fp32 matrix sits
// in LDS (pitch dim + 8 floats: a ds_read_b128 serves 8 rows x 2 k-quads per cycle, and + 32 B per row spreads those over all 64
// banks -- with + 16 B, as this kernel had it until round 3, 42 % of its LDS cycles were bank conflicts: profiles/r03_ai_*), every wave
// takes candidates round robin: per 16-row tile of the chunk and 16-wide k step one 16-B global load per lane (16 rows x 64 B: every byte
// of a row fetched once) feeds four v_mfma_f32_16x16x4_f32 steps per 16-column half of the query -- exact fp32 products,
// fp32 accumulation -- then max over the chunk's rows (rows past its end masked), sum over the query vectors in a fixed
// order.  Candidates < 0 (padding, sanitised ordinals) and empty chunks score -inf.
// Software pipeline (round 3): a wave is alone with one partner on its SIMD (the query fills the CU's LDS), so nothing hides a load it
// waits for.  The rows of the NEXT block of 128 k (the next tile's first, the next candidate's first) are requested before the 64 MFMAs
// of the current one, and the next candidate's ordinal and row range -- two dependent loads -- a whole candidate ahead: the counters
// showed the waves waiting on memory for 41 % of their cycles with the matrix pipe 46 % busy.
// KB: k per block (NL = KB / 16 loads per lane in flight ahead of their MFMAs: 256 k = 128 MFMAs = ~2 us of matrix pipe, about the
// latency of an HBM load under load; 128 left a third of it exposed);  FULL: dim % KB == 0 (no partial block: no branches in a block)
// ROW16: the rows are stored as fp16 (`D` points at halves): 8 B per lane and k step, converted to fp32 on the way in -- the stored
// halves ARE the corpus, the products are as exact as over an fp32 corpus
template <bool FULL, int KB, bool ROW16 = false>
__global__ __launch_bounds__(512) void maxsim_pairs_kernel(const float* __restrict__ D, int dim, const float* __restrict__ Q, int nq,
                                                            int64_t q_stride, const int64_t* __restrict__ offsets,
                                                            const int32_t* __restrict__ candidates, int64_t n_items, int64_t item_stride,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [32][dim + 8]
    const int pitch = dim + 8;
    const float* Qb = Q + (int64_t)blockIdx.y * q_stride;
    const int32_t* cb = candidates + (int64_t)blockIdx.y * item_stride;
    float* ob = out + (int64_t)blockIdx.y * item_stride;
    for (int i = threadIdx.x * 4; i < 32 * dim; i += 512 * 4) {
        const int n = i / dim, c = i - n * dim;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};  // query vectors past nq: zeros (their maxima are 0 and add nothing)
        if (n < nq) v = *reinterpret_cast<const f32x4*>(Qb + (int64_t)n * dim + c);
        *reinterpret_cast<f32x4*>(qs + n * pitch + c) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = wave_id();
    const int m = lane & 15, g = lane >> 4;  // A: row m of the tile, k quad g;  B / C: query column m, k quad / row quad g
    const float* q0 = qs + m * pitch + 4 * g;
    const float* q1 = qs + (16 + m) * pitch + 4 * g;
    const int64_t stride = (int64_t)gridDim.x * 8;
    // the 16 rows x 128 k of a block, one 16-B load per lane and 16-wide k step (rows past the chunk: its last row again, masked later)
    constexpr int NL = KB / 16;
    auto request = [&](f32x4 (&x)[NL], int64_t e, int64_t r0, int t0) __attribute__((always_inline)) {
        const int64_t row = r0 + m < e ? r0 + m : e - 1;
        if constexpr (ROW16) {
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            const _Float16* a = reinterpret_cast<const _Float16*>(D) + row * (int64_t)dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                h16x4 h = (h16x4){0, 0, 0, 0};
                if (FULL || t0 + 16 * j < dim) h = *reinterpret_cast<const h16x4*>(a + 16 * j);
                x[j] = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
        } else {
            const float* a = D + row * (int64_t)dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) x[j] = (FULL || t0 + 16 * j < dim) ? *reinterpret_cast<const f32x4*>(a + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto lane_i64 = [](int64_t v, int k) __attribute__((always_inline)) {  // lane k's value, k wave-uniform
        const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)v, k), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), k);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    f32x4 xc[NL], xn[NL];
    // 64 of this wave's candidates at a time: lane L looks up the L-th one's row range (two dependent loads, once per 64 candidates and
    // for all of them at the same time), candidates without rows get their -inf at once, the others are walked in lane order
    for (int64_t base = (int64_t)blockIdx.x * 8 + w; base < n_items; base += 64 * stride) {
        const int64_t mine = base + (int64_t)lane * stride;
        int64_t vb = 0, ve = 0;
        if (mine < n_items) {
            const int64_t chunk = cb[mine];
            if (chunk >= 0) { vb = offsets[chunk]; ve = offsets[chunk + 1]; }
            if (ve <= vb) ob[mine] = -INFINITY;
        }
        const uint64_t todo = __builtin_amdgcn_ballot_w64(ve > vb);
        if (todo == 0ull) continue;  // (wave-uniform)
        int k = __builtin_ctzll(todo);
        int64_t e = lane_i64(ve, k), r0 = lane_i64(vb, k);
        int t0 = 0;
        request(xc, e, r0, 0);
        f32x4 y0 = *reinterpret_cast<const f32x4*>(q0), y1 = *reinterpret_cast<const f32x4*>(q1);  // query fragments of the step being multiplied
        float best0 = -INFINITY, best1 = -INFINITY;  // running max over the chunk's rows of this lane's column (both halves)
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (;;) {
            // the block after this one -- same tile, next tile, the next candidate's first -- chosen without a branch and requested before
            // this block's 8 NL MFMAs (after the last block of the 64 candidates: the current candidate's first block again, unused)
            const bool more_k = t0 + KB < dim, more_rows = r0 + 16 < e;
            const uint64_t rest = todo & ~((2ull << k) - 1ull);
            const int nk = rest ? __builtin_ctzll(rest) : k;
            const int64_t ne = lane_i64(ve, nk), nb = lane_i64(vb, nk);
            const int64_t n_e = (more_k || more_rows) ? e : ne;
            const int64_t n_r0 = more_k ? r0 : (more_rows ? r0 + 16 : nb);
            const int n_t0 = more_k ? t0 + KB : 0;
            request(xn, n_e, n_r0, n_t0);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads below the MFMAs: they have to be in flight DURING them)
            // the query fragments of step j + 1 (of the next block's first step after the last) are read from LDS before the eight MFMAs
            // of step j: left to the compiler the reads sat right in front of their MFMAs, an LDS round trip exposed every 256 cycles
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                if (!FULL && t0 + 16 * j >= dim) break;  // (uniform)
                const int tn = j + 1 < NL ? t0 + 16 * (j + 1) : n_t0;  // (a partial block's steps past dim: a fragment nobody multiplies)
                const int tr = (FULL || tn < dim) ? tn : 0;
                const f32x4 z0 = *reinterpret_cast<const f32x4*>(q0 + tr);
                const f32x4 z1 = *reinterpret_cast<const f32x4*>(q1 + tr);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y0[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y1[u], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                y0 = z0;
                y1 = z1;
            }
            bool last = false;
            if (!more_k) {  // the tile is complete.  C layout: this lane holds rows 4 g + i (i = 0..3) of column m
                float v0 = -INFINITY, v1 = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + 4 * g + i < e) { v0 = fmaxf(v0, acc0[i]); v1 = fmaxf(v1, acc1[i]); }
                v0 = fmaxf(v0, __shfl_xor(v0, 16)); v0 = fmaxf(v0, __shfl_xor(v0, 32));
                v1 = fmaxf(v1, __shfl_xor(v1, 16)); v1 = fmaxf(v1, __shfl_xor(v1, 32));
                best0 = fmaxf(best0, v0);
                best1 = fmaxf(best1, v1);
                acc0 = acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!more_rows) {  // ... and so is the candidate: sum over the 32 query vectors, 16 columns per half by a 4-step butterfly,
                                   // then the two halves -- a fixed order
                    float s0 = best0, s1 = best1;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
                    if (lane == 0) ob[base + (int64_t)k * stride] = s0 + s1;
                    best0 = best1 = -INFINITY;
                    last = rest == 0ull;
                    k = nk;
                }
            }
            if (last) break;
            e = n_e;
            r0 = n_r0;
            t0 = n_t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) xc[j] = xn[j];
        }
    }
}

// maxsim_pairs_packed_kernel (round 4): the same scores, bit for bit, at about half the matrix work when the chunks are short.
// maxsim_pairs_kernel gives every candidate chunk its own 16-row MFMA tiles: at 1-15 rows per chunk (the RAGLite shape) half of every tile
// is padding, and the kernel is bound by the fp32 matrix pipe (512 v_mfma_f32_16x16x4_f32 per tile at dim 1024: 2 x 0.18 ms of every
// 128-query step of the headline pipeline).  Here a wave PACKS the rows of its (up to 64) candidates back to back into tiles: slot m of a
// tile is the next row of the current candidate, whichever that is -- a scalar walk over the candidates' row ranges (they sit in lanes:
// v_readlane) hands every lane its row and leaves a 16-bit mask of the slots that end a candidate; the tile epilogue takes the maximum per
// SEGMENT of slots (a candidate that straddles two tiles carries its column maxima over) and sums the 32 query vectors in the order of
// maxsim_pairs_kernel.  Per (row, query vector) the k steps accumulate in the same order whatever slot the row sits in, the maximum is
// exact, the sum tree is the same: identical bits (tests/test_gpu_pairs_packed.py).  dim % KB == 0 only (the other shapes keep the kernel above).
// DBG (experiment builds only, WRONG results): 1 = no MFMAs (the loads and the walk alone), 2 = the rows are loaded once per batch (the matrix
// work and the walk alone), 4 = no query staging
// NW: waves per workgroup (8; 16 -- option pairs_packed = 2, with KB = 128 so that a wave fits 128 VGPRs -- puts four waves on every SIMD
// instead of two: the same bytes in flight per CU, twice the waves to fill the matrix pipe while others wait for their rows)
template <int KB, bool ROW16, int DBG = 0, int NW = 8>
__global__ __launch_bounds__(NW * 64) void maxsim_pairs_packed_kernel(const float* __restrict__ D, int dim, const float* __restrict__ Q, int nq,
                                                                   int64_t q_stride, const int64_t* __restrict__ offsets,
                                                                   const int32_t* __restrict__ candidates, int64_t n_items, int64_t item_stride,
                                                                   float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [32][dim + 8]
    const int pitch = dim + 8;
    const float* Qb = Q + (int64_t)blockIdx.y * q_stride;
    const int32_t* cb = candidates + (int64_t)blockIdx.y * item_stride;
    float* ob = out + (int64_t)blockIdx.y * item_stride;
    if constexpr ((DBG & 16) != 0) return;  // (timing: the launch alone)
    for (int i = threadIdx.x * 4; i < 32 * dim && !(DBG & 4); i += NW * 64 * 4) {
        const int n = i / dim, c = i - n * dim;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (n < nq) v = *reinterpret_cast<const f32x4*>(Qb + (int64_t)n * dim + c);
        *reinterpret_cast<f32x4*>(qs + n * pitch + c) = v;
    }
    __syncthreads();
    if constexpr ((DBG & 8) != 0) return;  // (timing: launch + query staging)
    const int lane = threadIdx.x & 63, w = wave_id();
    const int m = lane & 15, g = lane >> 4;
    const float* q0 = qs + m * pitch + 4 * g;
    const float* q1 = qs + (16 + m) * pitch + 4 * g;
    const int64_t stride = (int64_t)gridDim.x * NW;
    constexpr int NL = KB / 16;
    auto request = [&](f32x4 (&x)[NL], int32_t row, int t0) __attribute__((always_inline)) {
        if constexpr ((DBG & 2) != 0) row = m;  // (timing: always the same sixteen rows, cache-resident)
        if constexpr (ROW16) {
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            const _Float16* a = reinterpret_cast<const _Float16*>(D) + (int64_t)row * dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const h16x4 h = *reinterpret_cast<const h16x4*>(a + 16 * j);
                x[j] = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
        } else {
            const float* a = D + (int64_t)row * dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) x[j] = *reinterpret_cast<const f32x4*>(a + 16 * j);
        }
    };
    f32x4 xc[NL], xn[NL];
    for (int64_t base = (int64_t)blockIdx.x * NW + w; base < n_items; base += 64 * stride) {
        const int64_t mine = base + (int64_t)lane * stride;
        int32_t vb = 0, ve = 0;  // (row numbers fit 31 bits: rl_index_create)
        if (mine < n_items) {
            const int64_t chunk = cb[mine];
            if (chunk >= 0) { vb = (int32_t)offsets[chunk]; ve = (int32_t)offsets[chunk + 1]; }
            if (ve <= vb) ob[mine] = -INFINITY;
        }
        const uint64_t todo = __builtin_amdgcn_ballot_w64(ve > vb);
        if (todo == 0ull) continue;  // (wave-uniform)
        // rows of all this wave's candidates (the same long chunk may be listed 64 times: the sum is taken in 32-bit halves, exactly)
        uint32_t t_lo = ve > vb ? (uint32_t)(ve - vb) & 0xffffu : 0u, t_hi = ve > vb ? (uint32_t)(ve - vb) >> 16 : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_lo += __shfl_xor(t_lo, o); t_hi += __shfl_xor(t_hi, o); }
        const int64_t R = ((int64_t)__builtin_amdgcn_readfirstlane(t_hi) << 16) + (int64_t)__builtin_amdgcn_readfirstlane(t_lo);
        const int64_t n_tiles = (R + 15) >> 4;
        // ---- the walk: candidate `cur` (a lane number), its next row, its end, the candidates still to come -- all scalar ----
        int cur = __builtin_ctzll(todo);
        uint64_t rest = todo & (todo - 1ull);
        int32_t cur_row = __builtin_amdgcn_readlane(vb, cur), cur_end = __builtin_amdgcn_readlane(ve, cur);
        // one tile: lane (m, g) gets the row of slot m (slots past the last row: the tile's first row, never read back), `cand` the lane
        // number of the slot's candidate; returns the mask of the slots that END a candidate
        auto next_tile = [&](int32_t& row, int& cand) __attribute__((always_inline)) -> uint32_t {
            uint32_t ends = 0u;
            row = cur_row;
            cand = cur;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                if (cur >= 0) {  // (scalar)
                    if (m == sl) { row = cur_row; cand = cur; }
                    if (++cur_row == cur_end) {
                        ends |= 1u << sl;
                        if (rest != 0ull) {
                            cur = __builtin_ctzll(rest);
                            rest &= rest - 1ull;
                            cur_row = __builtin_amdgcn_readlane(vb, cur);
                            cur_end = __builtin_amdgcn_readlane(ve, cur);
                        } else {
                            cur = -1;
                        }
                    }
                }
            }
            return ends;
        };
        int32_t row_c, row_n = 0;
        int cand_c, cand_n = 0;
        uint32_t ends_c = next_tile(row_c, cand_c), ends_n = 0u;
        int64_t tile = 0;
        int t0 = 0;
        request(xc, row_c, 0);
        f32x4 y0 = *reinterpret_cast<const f32x4*>(q0), y1 = *reinterpret_cast<const f32x4*>(q1);
        float carry0 = -INFINITY, carry1 = -INFINITY;  // column maxima of the candidate left open by the previous tile
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        for (;;) {
            const bool more_k = t0 + KB < dim, more_tiles = tile + 1 < n_tiles;  // (scalar)
            if (!more_k && more_tiles) ends_n = next_tile(row_n, cand_n);
            const int32_t n_row = (more_k || !more_tiles) ? row_c : row_n;
            const int n_t0 = more_k ? t0 + KB : 0;
            request(xn, n_row, n_t0);  // (after the last block of the batch: the current tile's first block again, unused)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int tn = j + 1 < NL ? t0 + 16 * (j + 1) : n_t0;
                const f32x4 z0 = *reinterpret_cast<const f32x4*>(q0 + tn);
                const f32x4 z1 = *reinterpret_cast<const f32x4*>(q1 + tn);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((DBG & 1) != 0) {  // (timing: no matrix work -- the loaded values must stay live)
                    acc0 += xc[j] * y0;
                    acc1 += xc[j] * y1;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y0[u], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y1[u], acc1, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                y0 = z0;
                y1 = z1;
            }
            if (!more_k) {  // the tile is complete: this lane holds rows (slots) 4 g + i, i = 0..3, of column m
                const int n_valid = R - 16 * tile < 16 ? (int)(R - 16 * tile) : 16;
                uint32_t em = ends_c;
                int s_lo = 0;
                while (s_lo < n_valid) {  // (scalar) one trip per candidate segment of the tile
                    const bool closes = em != 0u;
                    const int e = closes ? __builtin_ctz(em) : 15;  // (no end left: the candidate runs on into the next tile)
                    float v0 = -INFINITY, v1 = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (4 * g + i >= s_lo && 4 * g + i <= e) { v0 = fmaxf(v0, acc0[i]); v1 = fmaxf(v1, acc1[i]); }
                    v0 = fmaxf(v0, __shfl_xor(v0, 16)); v0 = fmaxf(v0, __shfl_xor(v0, 32));
                    v1 = fmaxf(v1, __shfl_xor(v1, 16)); v1 = fmaxf(v1, __shfl_xor(v1, 32));
                    const float best0 = fmaxf(carry0, v0), best1 = fmaxf(carry1, v1);
                    if (closes) {  // sum over the 32 query vectors: 16 columns per half by a 4-step butterfly, then the two halves
                        float s0 = best0, s1 = best1;
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
                        const int c = __builtin_amdgcn_readlane(cand_c, e);  // (lane e = slot e of row group 0)
                        if (lane == 0) ob[base + (int64_t)c * stride] = s0 + s1;
                        carry0 = carry1 = -INFINITY;
                        em &= em - 1u;
                    } else {
                        carry0 = best0;
                        carry1 = best1;
                    }
                    s_lo = e + 1;
                }
                acc0 = acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!more_tiles) break;
                ++tile;
                row_c = row_n;
                cand_c = cand_n;
                ends_c = ends_n;
            }
            t0 = n_t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) xc[j] = xn[j];
        }
    }
}

// maxsim_pairs_wide_kernel (round 6): the same scores for 1024 < dim <= 4096 (dim % 128 == 0) -- embedders wider than bge-m3 (the reference
// takes any litellm embedder, src/raglite/_embed.py:155-158: 1536- and 3072-wide models).  A query's [32 x dim] fp32 matrix no longer fits a CU's
// LDS, and a window of it SHARED by the workgroup would put a barrier between every wave's K blocks.  So the window is WAVE-PRIVATE: each of the
// eight waves keeps 32 x 128 floats of the query (17 KiB at the pitch of the kernels above) and turns the loops inside out -- a GROUP of up to
// eight packed tiles (the walk of maxsim_pairs_packed_kernel: the rows of the wave's candidates back to back) is multiplied window by window,
// its 8 x 2 accumulator tiles staying in registers, so that a window is staged once per 128 rows x 128 k (16 KiB from L2 against 64 KiB of rows
// from HBM); no workgroup barrier at all.  Per (row, query vector) the k steps accumulate in ascending order as in the kernels above, the
// maximum is exact, the sum tree is the same: the bits of maxsim_pairs_kernel's arithmetic at any dim (tests/test_gpu_wide_dim.py).
constexpr int PW_KW = 128, PW_T = 8, PW_PITCH = PW_KW + 8;
// ROW16: the rows are stored as fp16 (`D` points at halves; an fp16-stored wide index): 8 B per lane and k step, widened on the way in (exact)
template <bool ROW16>
__global__ __launch_bounds__(512) void maxsim_pairs_wide_kernel(const float* __restrict__ D, int dim, const float* __restrict__ Q, int nq,
                                                                 int64_t q_stride, const int64_t* __restrict__ offsets,
                                                                 const int32_t* __restrict__ candidates, int64_t n_items, int64_t item_stride,
                                                                 float* __restrict__ out, const uint32_t* __restrict__ run_if) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [8 waves][32][PW_PITCH]
    // run_if: a guarded launch (the full-precision fallback of a MaxSim batch over a wide index that keeps no pre-split image: EVERY chunk
    // is a candidate -- candidates == nullptr: item i is chunk i -- scored exactly; it returns at once unless the batch's flag is up)
    if (run_if && __builtin_amdgcn_readfirstlane((int)*run_if) == 0) return;
    const float* Qb = Q + (int64_t)blockIdx.y * q_stride;
    const int32_t* cb = candidates ? candidates + (int64_t)blockIdx.y * item_stride : nullptr;
    float* ob = out + (int64_t)blockIdx.y * item_stride;
    const int lane = threadIdx.x & 63, w = wave_id();
    const int m = lane & 15, g = lane >> 4;
    float* const qw = qs + w * 32 * PW_PITCH;
    const float* q0 = qw + m * PW_PITCH + 4 * g;
    const float* q1 = qw + (16 + m) * PW_PITCH + 4 * g;
    const int64_t stride = (int64_t)gridDim.x * 8;
    const int nwin = dim / PW_KW;
    constexpr int NL = PW_KW / 16;
    auto request = [&](f32x4 (&x)[NL], int32_t row, int t0) __attribute__((always_inline)) {
        if constexpr (ROW16) {
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            const _Float16* a = reinterpret_cast<const _Float16*>(D) + (int64_t)row * dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const h16x4 h = *reinterpret_cast<const h16x4*>(a + 16 * j);
                x[j] = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            }
        } else {
            const float* a = D + (int64_t)row * dim + 4 * g + t0;
#pragma unroll
            for (int j = 0; j < NL; ++j) x[j] = *reinterpret_cast<const f32x4*>(a + 16 * j);
        }
    };
    f32x4 xc[NL], xn[NL];
    for (int64_t base = (int64_t)blockIdx.x * 8 + w; base < n_items; base += 64 * stride) {
        const int64_t mine = base + (int64_t)lane * stride;
        int32_t vb = 0, ve = 0;  // (row numbers fit 31 bits: rl_index_create)
        if (mine < n_items) {
            const int64_t chunk = cb ? (int64_t)cb[mine] : mine;
            if (chunk >= 0) { vb = (int32_t)offsets[chunk]; ve = (int32_t)offsets[chunk + 1]; }
            if (ve <= vb) ob[mine] = -INFINITY;
        }
        const uint64_t todo = __builtin_amdgcn_ballot_w64(ve > vb);
        if (todo == 0ull) continue;  // (wave-uniform)
        uint32_t t_lo = ve > vb ? (uint32_t)(ve - vb) & 0xffffu : 0u, t_hi = ve > vb ? (uint32_t)(ve - vb) >> 16 : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_lo += __shfl_xor(t_lo, o); t_hi += __shfl_xor(t_hi, o); }
        const int64_t R = ((int64_t)__builtin_amdgcn_readfirstlane(t_hi) << 16) + (int64_t)__builtin_amdgcn_readfirstlane(t_lo);
        const int64_t n_tiles = (R + 15) >> 4;
        // the walk of maxsim_pairs_packed_kernel (scalar): slot m of a tile is the next row of the current candidate
        int cur = __builtin_ctzll(todo);
        uint64_t rest = todo & (todo - 1ull);
        int32_t cur_row = __builtin_amdgcn_readlane(vb, cur), cur_end = __builtin_amdgcn_readlane(ve, cur);
        auto next_tile = [&](int32_t& row, int& cand) __attribute__((always_inline)) -> uint32_t {
            uint32_t ends = 0u;
            row = cur_row;
            cand = cur;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                if (cur >= 0) {  // (scalar)
                    if (m == sl) { row = cur_row; cand = cur; }
                    if (++cur_row == cur_end) {
                        ends |= 1u << sl;
                        if (rest != 0ull) {
                            cur = __builtin_ctzll(rest);
                            rest &= rest - 1ull;
                            cur_row = __builtin_amdgcn_readlane(vb, cur);
                            cur_end = __builtin_amdgcn_readlane(ve, cur);
                        } else {
                            cur = -1;
                        }
                    }
                }
            }
            return ends;
        };
        float carry0 = -INFINITY, carry1 = -INFINITY;  // column maxima of the candidate left open by the previous tile
        for (int64_t tile0 = 0; tile0 < n_tiles; tile0 += PW_T) {
            const int gt = (int)(n_tiles - tile0 < PW_T ? n_tiles - tile0 : PW_T);  // tiles of this group (scalar)
            int32_t row[PW_T];
            int cand[PW_T];
            uint32_t ends[PW_T];
#pragma unroll
            for (int t = 0; t < PW_T; ++t) {
                row[t] = 0; cand[t] = 0; ends[t] = 0u;
                if (t < gt) ends[t] = next_tile(row[t], cand[t]);
            }
            f32x4 acc[PW_T][2];
#pragma unroll
            for (int t = 0; t < PW_T; ++t) acc[t][0] = acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            request(xc, row[0], 0);
            for (int kw = 0; kw < nwin; ++kw) {
                // this wave's window of the query: columns [128 kw, 128 kw + 128) of all 32 vectors (vectors past nq: zeros)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = j * 64 + lane, n = i >> 5, c = (i & 31) * 4;
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (n < nq) v = *reinterpret_cast<const f32x4*>(Qb + (int64_t)n * dim + kw * PW_KW + c);
                    *reinterpret_cast<f32x4*>(qw + n * PW_PITCH + c) = v;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < PW_T; ++t) {
                    if (t < gt) {  // (scalar)
                        const bool last_t = t + 1 >= gt;
                        const int32_t n_row = (t + 1 < PW_T && !last_t) ? row[t + 1 < PW_T ? t + 1 : 0] : row[0];
                        const int n_t0 = last_t ? (kw + 1 < nwin ? (kw + 1) * PW_KW : 0) : kw * PW_KW;
                        request(xn, n_row, n_t0);  // (after the group's last block: its first block again, unused)
                        f32x4 y0 = *reinterpret_cast<const f32x4*>(q0), y1 = *reinterpret_cast<const f32x4*>(q1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < NL; ++j) {
                            const int tn = j + 1 < NL ? 16 * (j + 1) : 0;
                            const f32x4 z0 = *reinterpret_cast<const f32x4*>(q0 + tn);
                            const f32x4 z1 = *reinterpret_cast<const f32x4*>(q1 + tn);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y0[u], acc[t][0], 0, 0, 0);
                                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[j][u], y1[u], acc[t][1], 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            y0 = z0;
                            y1 = z1;
                        }
#pragma unroll
                        for (int j = 0; j < NL; ++j) xc[j] = xn[j];
                    }
                }
            }
            // the group's tiles are complete: per tile the segment maxima and the sums of maxsim_pairs_packed_kernel, in tile order
#pragma unroll
            for (int t = 0; t < PW_T; ++t) {
                if (t < gt) {
                    const int64_t tile = tile0 + t;
                    const int n_valid = R - 16 * tile < 16 ? (int)(R - 16 * tile) : 16;
                    uint32_t em = ends[t];
                    int s_lo = 0;
                    while (s_lo < n_valid) {  // (scalar) one trip per candidate segment of the tile
                        const bool closes = em != 0u;
                        const int e = closes ? __builtin_ctz(em) : 15;
                        float v0 = -INFINITY, v1 = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (4 * g + i >= s_lo && 4 * g + i <= e) { v0 = fmaxf(v0, acc[t][0][i]); v1 = fmaxf(v1, acc[t][1][i]); }
                        v0 = fmaxf(v0, __shfl_xor(v0, 16)); v0 = fmaxf(v0, __shfl_xor(v0, 32));
                        v1 = fmaxf(v1, __shfl_xor(v1, 16)); v1 = fmaxf(v1, __shfl_xor(v1, 32));
                        const float best0 = fmaxf(carry0, v0), best1 = fmaxf(carry1, v1);
                        if (closes) {
                            float s0 = best0, s1 = best1;
#pragma unroll
                            for (int o = 1; o < 16; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
                            const int c = __builtin_amdgcn_readlane(cand[t], e);  // (lane e = slot e of row group 0)
                            if (lane == 0) ob[base + (int64_t)c * stride] = s0 + s1;
                            carry0 = carry1 = -INFINITY;
                            em &= em - 1u;
                        } else {
                            carry0 = best0;
                            carry1 = best1;
                        }
                        s_lo = e + 1;
                    }
                }
            }
        }
    }
}

int launch_maxsim_pairs(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride, const int64_t* offsets,
                        const int32_t* candidates, int64_t n_items_per_query, int32_t n_queries, float* out, hipStream_t s, bool rows16,
                        int64_t item_stride, int64_t first_item, int packed_mode) {
    bool packed = packed_mode != 0;
    if (n_items_per_query <= 0 || n_queries <= 0) return RL_OK;
    if (item_stride <= 0) item_stride = n_items_per_query;
    if (first_item < 0 || first_item + n_items_per_query > item_stride) return RL_ERR_INVALID;
    if (candidates) candidates += first_item;
    if (out) out += first_item;
    if (nq < 1 || nq > 32 || dim % 16 || dim < 16 || dim > PAIRS_MAX_DIM || !candidates) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & (rows16 ? 7 : 15)) || (reinterpret_cast<uintptr_t>(Q) & 15) || (q_stride & 3)) return RL_ERR_UNSUPPORTED;
    if (dim > 1024) {  // wider than a CU's LDS holds a query: wave-private query windows (maxsim_pairs_wide_kernel)
        if (dim % PW_KW) return RL_ERR_UNSUPPORTED;
        const size_t lds_w = (size_t)8 * 32 * PW_PITCH * sizeof(float);
        static bool wide_attr = false;
        if (!wide_attr) {
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            wide_attr = true;
        }
        const int per_q = (int)std::max<int64_t>(1, std::min<int64_t>((n_items_per_query + 7) / 8, std::max<int64_t>(1, 256 / n_queries)));
        if (rows16)
            hipLaunchKernelGGL(maxsim_pairs_wide_kernel<true>, dim3(per_q, n_queries), dim3(512), lds_w, s, D, (int)dim, Q, (int)nq, q_stride, offsets,
                               candidates, n_items_per_query, item_stride, out, (const uint32_t*)nullptr);
        else
            hipLaunchKernelGGL(maxsim_pairs_wide_kernel<false>, dim3(per_q, n_queries), dim3(512), lds_w, s, D, (int)dim, Q, (int)nq, q_stride, offsets,
                               candidates, n_items_per_query, item_stride, out, (const uint32_t*)nullptr);
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    const size_t lds = (size_t)32 * (dim + 8) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in
#define RL_PAIRS_ATTR(...) RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
#define RL_PACKED_ATTR(...) RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        RL_PACKED_ATTR(256, false);
        RL_PACKED_ATTR(128, false);
        RL_PACKED_ATTR(256, true);
        RL_PACKED_ATTR(128, true);
        RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<128, false, 0, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#undef RL_PACKED_ATTR
        RL_PAIRS_ATTR(true, 256, false);
        RL_PAIRS_ATTR(true, 128, false);
        RL_PAIRS_ATTR(false, 128, false);
        RL_PAIRS_ATTR(true, 256, true);
        RL_PAIRS_ATTR(true, 128, true);
        RL_PAIRS_ATTR(false, 128, true);
#undef RL_PAIRS_ATTR
        attr_set = true;
    }
    // workgroups per query: enough to fill the chip when there are few queries, at most one wave per candidate
    // (the packing kernel wants MANY candidates per wave -- its padding is one partial tile per wave and batch: one workgroup per CU and round)
    const bool full = dim % 128 == 0;
    packed = packed && full;
    const bool wide = packed && packed_mode == 2 && !rows16;  // sixteen waves per workgroup (KB = 128)
    const int nw = wide ? 16 : 8;
    int per_query = (int)std::max<int64_t>(1, std::min<int64_t>((n_items_per_query + nw - 1) / nw, std::max<int64_t>(1, (packed ? 256 : 512) / n_queries)));
#ifdef RAGLITE_EXPERIMENTS  // A/B of the split of a query's list over workgroups (scripts/gpu_calls/)
    if (const char* e = exp_env("RAGLITE_PAIRS_WG_BUDGET")) {
        const int budget = std::atoi(e);
        if (budget > 0) per_query = (int)std::max<int64_t>(1, std::min<int64_t>((n_items_per_query + 7) / 8, std::max<int64_t>(1, budget / n_queries)));
    }
#endif
#define RL_PAIRS(FULL_, KB_)                                                                                                               \
    do {                                                                                                                                   \
        if (rows16)                                                                                                                        \
            hipLaunchKernelGGL((maxsim_pairs_kernel<FULL_, KB_, true>), dim3(per_query, n_queries), dim3(512), lds, s, D, (int)dim, Q, (int)nq, \
                               q_stride, offsets, candidates, n_items_per_query, item_stride, out);                                        \
        else                                                                                                                               \
            hipLaunchKernelGGL((maxsim_pairs_kernel<FULL_, KB_, false>), dim3(per_query, n_queries), dim3(512), lds, s, D, (int)dim, Q, (int)nq, \
                               q_stride, offsets, candidates, n_items_per_query, item_stride, out);                                        \
    } while (0)
#define RL_PACKED(KB_)                                                                                                                     \
    do {                                                                                                                                   \
        if (rows16)                                                                                                                        \
            hipLaunchKernelGGL((maxsim_pairs_packed_kernel<KB_, true>), dim3(per_query, n_queries), dim3(512), lds, s, D, (int)dim, Q, (int)nq, \
                               q_stride, offsets, candidates, n_items_per_query, item_stride, out);                                        \
        else                                                                                                                               \
            hipLaunchKernelGGL((maxsim_pairs_packed_kernel<KB_, false>), dim3(per_query, n_queries), dim3(512), lds, s, D, (int)dim, Q, (int)nq, \
                               q_stride, offsets, candidates, n_items_per_query, item_stride, out);                                        \
    } while (0)
#ifdef RAGLITE_EXPERIMENTS
    static const int pdbg = exp_env("RAGLITE_PAIRS_DBG") ? std::atoi(exp_env("RAGLITE_PAIRS_DBG")) : 0;
    if (packed && dim % 256 == 0 && !rows16 && pdbg) {
#define RL_PDBG(D_) hipLaunchKernelGGL((maxsim_pairs_packed_kernel<256, false, D_>), dim3(per_query, n_queries), dim3(512), lds, s, D, (int)dim, Q, (int)nq, \
                                       q_stride, offsets, candidates, n_items_per_query, item_stride, out)
        static bool dbg_attr = false;
        if (!dbg_attr) {
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_packed_kernel<256, false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            dbg_attr = true;
        }
        if (pdbg == 1) RL_PDBG(1); else if (pdbg == 2) RL_PDBG(2); else if (pdbg == 3) RL_PDBG(3); else if (pdbg == 4) RL_PDBG(4); else if (pdbg == 8) RL_PDBG(8);
        else if (pdbg == 16) RL_PDBG(16); else RL_PDBG(7);
#undef RL_PDBG
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
#endif
    if (wide)
        hipLaunchKernelGGL((maxsim_pairs_packed_kernel<128, false, 0, 16>), dim3(per_query, n_queries), dim3(1024), lds, s, D, (int)dim, Q, (int)nq, q_stride,
                           offsets, candidates, n_items_per_query, item_stride, out);
    else if (packed && dim % 256 == 0) RL_PACKED(256);
    else if (packed && dim % 128 == 0) RL_PACKED(128);
    else if (dim % 256 == 0) RL_PAIRS(true, 256);
    else if (dim % 128 == 0) RL_PAIRS(true, 128);
    else RL_PAIRS(false, 128);
#undef RL_PACKED
#undef RL_PAIRS
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Exact MaxSim scores of EVERY chunk for a batch of queries over a wide index (1024 < dim <= 4096, dim % 128 == 0), behind a run-if flag:
// out[q * out_stride + chunk].  The guarded full-precision fallback of the bound-filtered batch where the index keeps rows + HI image only.
int launch_maxsim_pairs_all_wide(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride, const int64_t* offsets, int64_t n_chunks,
                                 int32_t n_queries, float* out, int64_t out_stride, hipStream_t s, const uint32_t* run_if, bool rows16) {
    if (n_chunks <= 0 || n_queries <= 0) return RL_OK;
    if (nq < 1 || nq > 32 || dim <= 1024 || dim > PAIRS_MAX_DIM || dim % PW_KW) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & (rows16 ? 7 : 15)) || (reinterpret_cast<uintptr_t>(Q) & 15) || (q_stride & 3)) return RL_ERR_UNSUPPORTED;
    static bool attr = false;
    if (!attr) {
        RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxsim_pairs_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const size_t lds_w = (size_t)8 * 32 * PW_PITCH * sizeof(float);
    const int per_q = (int)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 511) / 512, std::max<int64_t>(1, 1024 / n_queries)));
    if (rows16)
        hipLaunchKernelGGL(maxsim_pairs_wide_kernel<true>, dim3(per_q, n_queries), dim3(512), lds_w, s, D, (int)dim, Q, (int)nq, q_stride, offsets,
                           (const int32_t*)nullptr, n_chunks, out_stride, out, run_if);
    else
    hipLaunchKernelGGL(maxsim_pairs_wide_kernel<false>, dim3(per_q, n_queries), dim3(512), lds_w, s, D, (int)dim, Q, (int)nq, q_stride, offsets,
                       (const int32_t*)nullptr, n_chunks, out_stride, out, run_if);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// One launch per query (the generic path is a correctness backstop, not a throughput path).
int launch_maxsim_generic(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride,
                          const int64_t* offsets, const int32_t* candidates, int64_t n_items_per_query,
                          int32_t n_queries, float* out, hipStream_t s) {
    if (nq < 1 || nq > GEN_NQ_MAX) return fail(RL_ERR_UNSUPPORTED, "MaxSim: nq must be in [1, 1024]");
    if (n_items_per_query <= 0 || n_queries <= 0) return RL_OK;
    const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(D) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(Q) & 15) == 0) && ((q_stride * 4) % 16 == 0);
    for (int32_t qi = 0; qi < n_queries; ++qi) {
        const float* Qq = Q + (int64_t)qi * q_stride;
        const int32_t* cq = candidates ? candidates + (int64_t)qi * n_items_per_query : nullptr;
        float* oq = out + (int64_t)qi * n_items_per_query;
        int st = RL_ERR_UNSUPPORTED;
        if (vec4) {
            const int nv = (dim + 255) / 256;
            if (nv <= 1) st = generic_t<1, 4>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 2) st = generic_t<2, 4>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 4) st = generic_t<4, 4>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 8) st = generic_t<8, 4>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 16) st = generic_t<16, 4>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
        } else {
            const int nv = (dim + 63) / 64;
            if (nv <= 2) st = generic_t<2, 1>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 8) st = generic_t<8, 1>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
            else if (nv <= 32) st = generic_t<32, 1>(D, dim, Qq, nq, offsets, cq, n_items_per_query, oq, s);
        }
        if (st == RL_ERR_UNSUPPORTED) return fail(RL_ERR_UNSUPPORTED, "MaxSim: dim too large for the generic kernel");
        RL_TRY(st);
    }
    return RL_OK;
}

// ---- rerank fast path: dim == 128 ----------------------------------------------------------------------
constexpr int CD = 128;   // embedding dimension of the rerank fast path
constexpr int CPW = 8;    // candidates per wave

// F16: D holds IEEE fp16 rows (fp16-stored index); the query is split into fp16 hi + lo halves for
// v_mfma_f32_16x16x32_f16 exactly as in maxsim_stream.hip.
// SPLIT: the fp16 (hi, lo) arithmetic of maxsim_stream.hip for an fp32-stored corpus (include/raglite_hip.h,
// RL_ARITH_F16_SPLIT): with exact fp32 MFMAs this shape sits right on the ridge (matrix pipe 62 %, HBM 76 %).
template <int NQT, bool F16, bool SPLIT = false>
__global__ __launch_bounds__(256) void maxsim_cand_kernel(const float* __restrict__ D, const float* __restrict__ Q,
                                                           int nq, const int64_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ candidates, int n_cand,
                                                           int n_queries, float* __restrict__ out, float e_scale) {
    static_assert(!(F16 && SPLIT), "SPLIT is a way to multiply an fp32-stored corpus");
    const int lane = threadIdx.x & 63;
    const int fj = lane & 15, kq = lane >> 4;
    const int groups = (n_cand + CPW - 1) / CPW;            // waves per query
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (int64_t)n_queries * groups) return;
    const int qi = (int)(wave / groups);
    const int c0 = (int)(wave % groups) * CPW;

    // B fragments: lane (j, kq), MFMA 4*mm + tt uses Q[16h + j][16*mm + 4*kq + tt]  (fp16 storage: MFMA mm uses
    // Q[16h + j][32*mm + 8*kq .. +7] as hi / lo halves)
    constexpr bool H = F16 || SPLIT;
    [[maybe_unused]] float qreg[H ? 1 : NQT][H ? 1 : 32];
    [[maybe_unused]] h16x8 qhi[H ? NQT : 1][H ? 4 : 1], qlo[H ? NQT : 1][H ? 4 : 1];
    [[maybe_unused]] bool any_lo = false;
    [[maybe_unused]] float q_unscale = 1.f;
    const float* Qq = Q + (int64_t)qi * nq * CD;
    if constexpr (SPLIT) {
        // the whole query (nq x 128) scaled to [2^13, 2^14) by one power of two, split into fp16 (hi, lo); MFMA m uses
        // k = 16 (2m + (u >> 2)) + 4 kq + (u & 3), the positions of A fragments 2m and 2m + 1
        float mxq = 0.f;
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
            const int qv = 16 * h + fj;
            if (qv < nq)
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(Qq + (int64_t)qv * CD + 16 * mm + 4 * kq);
                    mxq = fmaxf(mxq, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxq = fmaxf(mxq, __shfl_xor(mxq, o));
        int ex = 0;
        if (mxq > 0.f && mxq < INFINITY) (void)frexpf(mxq, &ex);
        const float q_scale = ldexpf(1.f, 14 - ex);
        q_unscale = ldexpf(1.f, ex - 14) / e_scale;
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
            const int qv = 16 * h + fj;
            const int qc_ = qv < nq ? qv : nq - 1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float* qp = Qq + (int64_t)qc_ * CD + 32 * m + 4 * kq;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp), v1 = *reinterpret_cast<const f32x4*>(qp + 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = qv < nq ? (u < 4 ? v0[u] : v1[u - 4]) * q_scale : 0.f;
                    const _Float16 hi = (_Float16)x;
                    const _Float16 lo = (_Float16)(x - (float)hi);
                    qhi[h][m][u] = hi;
                    qlo[h][m][u] = lo;
                    any_lo |= lo != (_Float16)0.0f;
                }
            }
        }
        any_lo = __builtin_amdgcn_ballot_w64(any_lo) != 0;
    }
#pragma unroll
    for (int h = 0; h < (SPLIT ? 0 : NQT); ++h) {
        const int qv = 16 * h + fj;
        const int qc_ = qv < nq ? qv : nq - 1;
        if constexpr (!F16) {
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Qq + (int64_t)qc_ * CD + 16 * mm + 4 * kq);
                if (qv >= nq) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                qreg[h][4 * mm + 0] = v[0]; qreg[h][4 * mm + 1] = v[1];
                qreg[h][4 * mm + 2] = v[2]; qreg[h][4 * mm + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const float* qp = Qq + (int64_t)qc_ * CD + 32 * mm + 8 * kq;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp), v1 = *reinterpret_cast<const f32x4*>(qp + 4);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = qv < nq ? (u < 4 ? v0[u] : v1[u - 4]) : 0.f;
                    const _Float16 hi = (_Float16)x;
                    const _Float16 lo = (_Float16)(x - (float)hi);
                    qhi[h][mm][u] = hi;
                    qlo[h][mm][u] = lo;
                    any_lo |= lo != (_Float16)0.0f;
                }
            }
        }
    }
    if constexpr (F16) any_lo = __builtin_amdgcn_ballot_w64(any_lo) != 0;

    for (int ci = c0; ci < c0 + CPW && ci < n_cand; ++ci) {
        const int64_t chunk = candidates[(int64_t)qi * n_cand + ci];
        // a negative ordinal = "no chunk" (padding of a search result, or sanitised by rl_maxsim_rerank): score -inf
        const int64_t b = chunk >= 0 ? offsets[chunk] : 0, e = chunk >= 0 ? offsets[chunk + 1] : 0;
        f32x4 mx[NQT];
#pragma unroll
        for (int h = 0; h < NQT; ++h) mx[h] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int64_t r0 = b; r0 < e; r0 += 16) {
            int64_t row = r0 + fj;            // A fragment: lane (i = fj, k = kq) supplies row i
            if (row > e - 1) row = e - 1;     // clamp; masked below
            f32x4 acc[NQT];
#pragma unroll
            for (int h = 0; h < NQT; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                const float* p = D + row * CD + 4 * kq;
                f32x4 a[8];
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) a[mm] = *reinterpret_cast<const f32x4*>(p + 16 * mm);
                f32x4 acl[NQT];
#pragma unroll
                for (int h = 0; h < NQT; ++h) acl[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    h16x8 eh, el;
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        const float x0 = a[2 * m + (u >> 2)][u & 3] * e_scale, x1 = a[2 * m + (u >> 2)][(u & 3) + 1] * e_scale;
                        const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);  // truncation: the residual is exact in fp32
                        const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
                        eh[u] = ph[0]; eh[u + 1] = ph[1];
                        el[u] = pl[0]; el[u + 1] = pl[1];
                    }
#pragma unroll
                    for (int h = 0; h < NQT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, qhi[h][m], acc[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < NQT; ++h) acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, qhi[h][m], acl[h], 0, 0, 0);
                    if (any_lo) {
#pragma unroll
                        for (int h = 0; h < NQT; ++h) acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, qlo[h][m], acl[h], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int h = 0; h < NQT; ++h) acc[h] = (acc[h] + acl[h]) * q_unscale;
            } else if constexpr (!F16) {
                const float* p = D + row * CD + 4 * kq;
                f32x4 a[8];
#pragma unroll
                for (int mm = 0; mm < 8; ++mm) a[mm] = *reinterpret_cast<const f32x4*>(p + 16 * mm);
#pragma unroll
                for (int mm = 0; mm < 8; ++mm)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int h = 0; h < NQT; ++h)
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mm][tt], qreg[h][4 * mm + tt], acc[h], 0, 0, 0);
            } else {
                const uint16_t* p = reinterpret_cast<const uint16_t*>(D) + row * CD + 8 * kq;
                h16x8 a[4];
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) a[mm] = *reinterpret_cast<const h16x8*>(p + 32 * mm);
#pragma unroll
                for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                    for (int h = 0; h < NQT; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mm], qhi[h][mm], acc[h], 0, 0, 0);
                if (any_lo) {
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                        for (int h = 0; h < NQT; ++h)
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mm], qlo[h][mm], acc[h], 0, 0, 0);
                }
            }
            // C/D: column fj = query, row 4*kq + reg = corpus row r0 + 4*kq + reg
#pragma unroll
            for (int h = 0; h < NQT; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = (r0 + 4 * kq + r) < e;
                    mx[h][r] = fmaxf(mx[h][r], valid ? acc[h][r] : -INFINITY);
                }
        }
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
            float v = fmaxf(fmaxf(mx[h][0], mx[h][1]), fmaxf(mx[h][2], mx[h][3]));
            v = fmaxf(v, __shfl_xor(v, 16, 64));
            v = fmaxf(v, __shfl_xor(v, 32, 64));
            s += (16 * h + fj < nq) ? v : 0.f;
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (lane == 0) out[(int64_t)qi * n_cand + ci] = (e > b) ? s : -INFINITY;
    }
}

static int launch_cand_any(const float* D, bool f16, int32_t dim, const float* Q, int32_t nq, const int64_t* offsets,
                           const int32_t* candidates, int32_t n_cand, int32_t n_queries, float* out, hipStream_t s,
                           float split_scale = 0.f) {
    if (dim != CD || nq < 1 || nq > 32) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    if (n_cand <= 0 || n_queries <= 0) return RL_OK;
    const int64_t waves = (int64_t)n_queries * ((n_cand + CPW - 1) / CPW);
    const int64_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "MaxSim rerank: too many (query, candidate) pairs per launch");
#define RL_CAND(NQT, ...) hipLaunchKernelGGL((maxsim_cand_kernel<NQT, __VA_ARGS__>), dim3((unsigned)blocks), dim3(256), 0, s, D, Q, \
                                             (int)nq, offsets, candidates, (int)n_cand, (int)n_queries, out, split_scale)
    if (f16) { if (nq <= 16) RL_CAND(1, true); else RL_CAND(2, true); }
    else if (split_scale > 0.f) { if (nq <= 16) RL_CAND(1, false, true); else RL_CAND(2, false, true); }
    else     { if (nq <= 16) RL_CAND(1, false); else RL_CAND(2, false); }
#undef RL_CAND
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Candidate ordinals that cannot be scored -- outside [0, n_chunks), or tombstoned -- become -1, which the kernels above
// score as -inf (rl_chunk_best_rows treats -1 the same way).
__global__ __launch_bounds__(256) void sanitize_candidates_kernel(const int32_t* __restrict__ in, int64_t n, int64_t n_chunks,
                                                                   const uint32_t* __restrict__ live_chunk_bits,
                                                                   int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t c = in[i];
    bool ok = c >= 0 && c < n_chunks;
    if (ok && live_chunk_bits) ok = (live_chunk_bits[c >> 5] >> (c & 31)) & 1u;
    out[i] = ok ? c : -1;
}
int launch_sanitize_candidates(const int32_t* in, int64_t n, int64_t n_chunks, const uint32_t* live_chunk_bits, int32_t* out,
                               hipStream_t s) {
    if (n <= 0) return RL_OK;
    hipLaunchKernelGGL(sanitize_candidates_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, n_chunks, live_chunk_bits, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_maxsim_cand(const float* D, int32_t dim, const float* Q, int32_t nq, const int64_t* offsets,
                       const int32_t* candidates, int32_t n_cand, int32_t n_queries, float* out, hipStream_t s,
                       float split_scale) {
    return launch_cand_any(D, false, dim, Q, nq, offsets, candidates, n_cand, n_queries, out, s, split_scale);
}

int launch_maxsim_cand16(const uint16_t* D, int32_t dim, const float* Q, int32_t nq, const int64_t* offsets,
                         const int32_t* candidates, int32_t n_cand, int32_t n_queries, float* out, hipStream_t s) {
    return launch_cand_any(reinterpret_cast<const float*>(D), true, dim, Q, nq, offsets, candidates, n_cand, n_queries,
                           out, s);
}

}  // namespace rl
