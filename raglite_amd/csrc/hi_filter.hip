// Small kernels of the half-bytes single-query search (api.hip: search_rows_hi): a corpus pass over the HI halves only
// (2 B per element) ranks approximately, a rigorous error bound turns the approximate top of the list into a candidate set
// that provably contains the exact top-k, and the candidates are re-scored with the exact kernels.
//
// a6 + a7 of SURVEY.md section 8a (src/raglite/_search.py:69-79) for B <= 16 queries: results identical to the full-precision
// pass, bit for bit.
#include "common.h"

namespace rl {
namespace {

// l2 (round 6).  The similarity is 1 - sqrt(D), D = |e - q|^2; the approximate one is made of D~ = |e|^2 + |q|^2 - 2 d~ (transform_score) with d~
// the dot product over the hi halves.  |D~ - D| <= delta for every row, D the squared distance AS THE EXACT KERNEL COMPUTES IT:
//   2 a |q|                      the dot's bound (a = max |e_lo| + eps max |e|: what the halves drop + the roundings of both dots),
//   eps (|e|^2 + |q|^2) / 2 ...  the fp32 sums behind |e|^2, |q|^2 and the exact sum (e - q)^2 (each off by at most dim 2^-24 of itself),
//   2^-22 (|e| + |q|)^2          the three fp32 operations of the transform
// <= 2 a |q| + 2 eps (max |e| + |q|)^2 with eps = 2^-12 per started 1024 terms (api.hip: sum_eps).  A row of the exact top-k has
// D <= D_k <= D~_k + delta (the k rows of the approximate top-k bound the exact k-th from above), so D~ <= D~_k + 2 delta: in similarities,
// with ANY lower bound P of the k-th best approximate one, s~ >= 1 - sqrt((1 - P)^2 + 2 delta) -- the slack factors cover the roundings of
// sqrt and 1 - x on both sides.
// (l2_delta / lower_threshold: common.h -- the ranked flow of select.hip uses them too)

// One block per query.  topk[b * k + j] = the k best APPROXIMATE similarities, descending.  With |approx - exact| <= m for every
// row, a row can be in the exact top-k only if its approximate score is >= (k-th best approximate) - 2 m =: thr[b]:
//   cosine: m = m_rel (the scores are cosines);  dot: m = m_rel * e_norm_bound * |q|.
// An unusable k-th score (NaN: fewer than k comparable rows) sets *flag: the caller's guarded full-precision pass then runs.
__global__ __launch_bounds__(256) void approx_threshold_kernel(const float* __restrict__ topk, int32_t nb, int32_t k,
                                                                const float* __restrict__ queries, int dim, int mode, float m_rel,
                                                                float e_norm_bound, float* __restrict__ thr, uint32_t* __restrict__ cnt,
                                                                uint32_t* __restrict__ flag, float e_max) {
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
            // (dot: the similarity is 1 + d, rounded to fp32 in both passes -- 2^-22 absolute covers that for scores up to 2)
            const float m = mode == SCAN_COSINE ? m_rel : mode == SCAN_L2 ? l2_delta(e_norm_bound, m_rel, e_max, qn) : m_rel * e_norm_bound * qn + 0x1p-22f;
            const float t = lower_threshold(topk[(int64_t)b * k + (k - 1)], m, mode == SCAN_L2);
            thr[b] = t;
            cnt[b] = 0u;
            bad |= !(t > -INFINITY);  // NaN or -inf
        }
    }
    if (threadIdx.x == 0) *flag = bad ? 1u : 0u;
}

// Rows whose approximate score reaches thr[b] -> ids[b * cap + p] (any order; with row_norm their norms next to them, for the
// cosine transform of the re-scored candidates), cnt[b] = how many; more than cap sets *flag.
// A workgroup gathers its hits in LDS first (one LDS atomic per wave and visit) and takes its range of the list with ONE global atomic:
// a MaxSim batch collects ~300 chunks for each of 128 queries, and 300 returning atomics on one word are 300 dependent L2 round trips
// -- the kernel took 0.16 ms for 64 MB of scores with one global atomic per hit-carrying wave (profiles/r03_ac_bench_kernel_stats.csv).
// More than LOCAL hits in one workgroup's ~4 k scores: the guarded full-precision path answers (*flag), as for a full list.
// top_s / top_i != nullptr (the second, tighter threshold of a MaxSim batch, api.hip: hi_batch_rescore): [nb x k] the approximate top-k
// in selection order (score desc, id asc) -- its members are already in the list (positions 0 .. k - 1, cnt[b] starts at k) and are NOT
// collected again: an entry is taken only if it ranks BELOW the k-th one, (key, id) < (key_k, id_k) in the selection's own order.
// (the body, shared with pivot_collect_kernel: every thread of the 256-thread block must call)
__device__ __forceinline__ void collect_rows(const float* __restrict__ s, int64_t n, bool vec, float t, const float* __restrict__ row_norm, int32_t cap,
                                             int32_t* __restrict__ ids, float* __restrict__ norms, uint32_t* __restrict__ cnt_b,
                                             uint32_t* __restrict__ flag, bool below_top, uint32_t key_k, int64_t id_k, int32_t* l_ids, uint32_t* l_n,
                                             uint32_t* l_base, const float* __restrict__ E = nullptr, int dim = 0, float* __restrict__ G = nullptr) {
    constexpr uint32_t LOCAL = 1024;
    if (threadIdx.x == 0) *l_n = 0u;
    __syncthreads();
    auto test = [&](float v, int64_t i, bool in_range) {
        bool hit = in_range && v >= t;
        if (below_top) {
            const uint32_t kv = score_key(v);
            hit = hit && (kv < key_k || (kv == key_k && i > id_k));
        }
        return hit;
    };
    auto append = [&](bool hit, int64_t i) {
        const uint64_t mask = __builtin_amdgcn_ballot_w64(hit);
        if (mask == 0ull) return;  // (wave-uniform)
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __builtin_ctzll(mask)) base = atomicAdd(l_n, (uint32_t)__builtin_popcountll(mask));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(mask));
        if (hit) {
            const uint32_t p = base + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (p < LOCAL) l_ids[p] = (int32_t)i;
        }
    };
    auto visit = [&](float v, int64_t i, bool in_range) { append(test(v, i, in_range), i); };
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (vec) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* s4 = reinterpret_cast<const f4*>(s);
        const int64_t n4 = n >> 2;
        // (whole waves walk the loop together -- the ballots need every lane -- so the trip count is rounded up per block; two 16-byte loads
        // in flight per lane, ONE ballot per group of four scores where nothing passes: the common case by four orders of magnitude)
        for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n4; i0 += 2 * stride) {
            const int64_t ia = i0 + threadIdx.x, ib = ia + stride;
            const bool in_a = ia < n4, in_b = ib < n4;
            const f4 va = in_a ? s4[ia] : (f4){0.f, 0.f, 0.f, 0.f};
            const f4 vb = in_b ? s4[ib] : (f4){0.f, 0.f, 0.f, 0.f};
            bool ha[4], hb[4], any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) { ha[u] = test(va[u], (ia << 2) + u, in_a); hb[u] = test(vb[u], (ib << 2) + u, in_b); any |= ha[u] | hb[u]; }
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) continue;  // (wave-uniform)
#pragma unroll
            for (int u = 0; u < 4; ++u) append(ha[u], (ia << 2) + u);
#pragma unroll
            for (int u = 0; u < 4; ++u) append(hb[u], (ib << 2) + u);
        }
        if (blockIdx.x == 0 && threadIdx.x < 64) {  // the n % 4 rows at the end: one wave
            const bool in = threadIdx.x < (n & 3);
            visit(in ? s[(n4 << 2) + threadIdx.x] : 0.f, (n4 << 2) + threadIdx.x, in);
        }
    } else {
        for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += stride) {
            const int64_t i = i0 + threadIdx.x;
            visit(i < n ? s[i] : 0.f, i, i < n);
        }
    }
    __syncthreads();
    const uint32_t found = *l_n;
    if (found == 0u) return;  // (workgroup-uniform)
    const uint32_t mine = found < LOCAL ? found : LOCAL;
    if (threadIdx.x == 0) {
        *l_base = atomicAdd(cnt_b, mine);
        if (found > LOCAL) atomicOr(flag, 1u);
    }
    __syncthreads();
    const uint32_t base = *l_base;
    for (uint32_t j = threadIdx.x; j < mine; j += 256) {
        const uint32_t p = base + j;
        if (p < (uint32_t)cap) {
            const int32_t i = l_ids[j];
            ids[p] = i;
            if (row_norm) norms[p] = row_norm[i];
        } else {
            atomicOr(flag, 1u);
        }
    }
    // E != nullptr: the rows themselves go to G[p] right away (dim % 4 == 0, 16-byte aligned: the launcher checks) -- the workgroup that found a
    // row copies it, a handful of rows per workgroup and all workgroups at once: no gather launch behind the collection
    if (E) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (uint32_t j = 0; j < mine && base + j < (uint32_t)cap; ++j) {  // (workgroup-uniform)
            const f4* src = reinterpret_cast<const f4*>(E + (int64_t)l_ids[j] * dim);
            f4* dst = reinterpret_cast<f4*>(G + (int64_t)(base + j) * dim);
            for (int c = threadIdx.x; c < (dim >> 2); c += 256) dst[c] = src[c];
        }
    }
}

__global__ __launch_bounds__(256) void collect_above_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, const float* __restrict__ thr,
                                                             const float* __restrict__ row_norm, int32_t cap, int32_t* __restrict__ ids,
                                                             float* __restrict__ norms, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag,
                                                             const float* __restrict__ top_s, const int32_t* __restrict__ top_i, int32_t k) {
    __shared__ int32_t l_ids[1024];
    __shared__ uint32_t l_n, l_base;
    const int b = blockIdx.y;
    const bool below_top = top_s != nullptr;
    const uint32_t key_k = below_top ? score_key(top_s[(int64_t)b * k + (k - 1)]) : 0u;
    const int64_t id_k = below_top ? (int64_t)top_i[(int64_t)b * k + (k - 1)] : -1;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
    collect_rows(scores + (int64_t)b * ld, n, vec, thr[b], row_norm, cap, ids + (int64_t)b * cap, norms ? norms + (int64_t)b * cap : nullptr, cnt + b, flag,
                 below_top, key_k, id_k, l_ids, &l_n, &l_base);
}

// ---- the single-query row search without an approximate RANKING (round 6, api.hip: search_rows_hi, option hi_pivot) ---------------------------
// What the half-bytes search needs from its approximate similarities is a candidate set, not their order: every row whose approximate
// similarity reaches (k-th best approximate) - 2 m.  ANY lower bound P of the k-th best gives a superset {approx >= P - 2 m}, and the
// candidates are re-scored and ranked exactly anyway.  P = the k-th largest of G >= 3 k workgroup MAXIMA (G distinct rows): about
// -G ln(1 - k / G) rows reach it (128 of a million at k = 100, G = 489) -- as good as the exact k-th best for this purpose, and two launches
// (transform + maxima; pivot + collection) instead of three (transform + histogram, filter, final: the radix selection of a million scores).
//
// (1) raw dots -> similarities in place (transform_score: the statements of transform_hist_kernel, same bits), the best key of every workgroup
// -> bmax[b * G + blockIdx.x]; the bound m_b of the query (HiBound) and the zeroed counter block, as transform_hist_kernel leaves them.
__global__ __launch_bounds__(256) void transform_bmax_kernel(float* __restrict__ scores, int64_t n, int64_t ld, const float* __restrict__ row_norm,
                                                              const float* __restrict__ row_sumsq, const float* __restrict__ queries, int dim,
                                                              int mode, float pre_scale, uint64_t* __restrict__ bmax, uint32_t* __restrict__ zero_words,
                                                              int n_zero, HiBound hb, PivotMaxSim ms) {
    __shared__ float part[4];
    __shared__ uint64_t wmax[4];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && b == 0 && (int)threadIdx.x < n_zero) zero_words[threadIdx.x] = 0u;
    // MaxSim flavour (ms.nq > 0; the scores are chunk scores, mode = raw): the bound is m_abs * sum_i |q_i| over the query's nq vectors (the
    // statement of maxsim_threshold_kernel: a wave per vector), computed by ONE workgroup per query AFTER its share of the scores (below);
    // ms.fill_ids: the candidate lists start out as "no chunk" (-1) everywhere
    if (ms.fill_ids)
        for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < ms.fill_n; i += (int64_t)gridDim.x * gridDim.y * 256)
            ms.fill_ids[i] = -1;
    float* const sb = scores + (int64_t)b * ld;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;  // (row_norm / row_sumsq are hipMalloc'd)
    const int64_t n4 = vec ? n >> 2 : 0;
    const float* aux = mode == SCAN_COSINE ? row_norm : mode == SCAN_L2 ? row_sumsq : nullptr;
    // the first trip's loads are issued before the query-norm reduction so that their latency hides behind it
    const int64_t i_first = (int64_t)blockIdx.x * 256 + threadIdx.x;
    f4 d0 = (f4){0.f, 0.f, 0.f, 0.f}, a0 = d0;
    if (i_first < n4) {
        d0 = reinterpret_cast<const f4*>(sb)[i_first];
        if (aux) a0 = reinterpret_cast<const f4*>(aux)[i_first];
    }
    float ss = 0.f;
    for (int c = threadIdx.x; c < dim && ms.nq == 0; c += 256) {
        const float v = queries[(int64_t)b * dim + c];
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float qss = (part[0] + part[1]) + (part[2] + part[3]);
    const float qn = sqrtf(qss);
    if (hb.m_out && ms.nq == 0 && blockIdx.x == 0 && threadIdx.x == 0)
        hb.m_out[b] = mode == SCAN_COSINE ? hb.m_rel : mode == SCAN_L2 ? l2_delta(hb.e_norm_bound, hb.m_rel, hb.e_max, qn) : hb.m_rel * hb.e_norm_bound * qn + 0x1p-22f;
    float bv = -INFINITY;  // ONE of this thread's best rows (the first of equal similarities; NaN / -inf never win)
    uint32_t bi = 0u;
    auto finish = [&](int64_t i, const f4 d, const f4 a) {
        f4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = transform_score(d[u] * pre_scale, mode, a[u], a[u], qn, qss);
            if (o[u] > bv) { bv = o[u]; bi = (uint32_t)((i << 2) + u); }
        }
        if (!ms.read_only) reinterpret_cast<f4*>(sb)[i] = o;
    };
    auto one = [&](int64_t i) {  // scalar tail / unaligned layout
        const float o = transform_score(sb[i] * pre_scale, mode, mode == SCAN_COSINE ? row_norm[i] : 1.f, mode == SCAN_L2 ? row_sumsq[i] : 0.f, qn, qss);
        if (o > bv) { bv = o; bi = (uint32_t)i; }
        if (!ms.read_only) sb[i] = o;
    };
    if (vec) {
        int64_t i = i_first;
        while (i < n4) {
            const int64_t nx = i + stride;
            f4 d1 = (f4){0.f, 0.f, 0.f, 0.f}, a1 = d1;
            if (nx < n4) {
                d1 = reinterpret_cast<const f4*>(sb)[nx];
                if (aux) a1 = reinterpret_cast<const f4*>(aux)[nx];
            }
            finish(i, d0, a0);
            d0 = d1; a0 = a1;
            i = nx;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) one((n4 << 2) + threadIdx.x);
    } else {
        for (int64_t i = i_first; i < n; i += stride) one(i);
    }
    uint64_t mine = bv > -INFINITY ? make_key64(bv, bi) : 0ull;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)mine, o, 64), hi = __shfl_xor((uint32_t)(mine >> 32), o, 64);
        const uint64_t other = ((uint64_t)hi << 32) | lo;
        mine = other > mine ? other : mine;
    }
    if (ms.per_wave) {  // few scores per query (MaxSim chunk scores): a maximum per WAVE -- 4 x gridDim.x groups
        if ((threadIdx.x & 63) == 0) bmax[((int64_t)b * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = mine;
    } else {
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t m01 = wmax[0] > wmax[1] ? wmax[0] : wmax[1], m23 = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
            bmax[(int64_t)b * gridDim.x + blockIdx.x] = m01 > m23 ? m01 : m23;
        }
    }
    if (ms.nq > 0 && hb.m_out && blockIdx.x == 0) {  // (workgroup-uniform)
        const float* Qb = queries + (int64_t)b * ms.q_stride;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float w_norms = 0.f;
        for (int i = wv; i < ms.nq; i += 4) {
            float s2 = 0.f;
            if ((dim & 3) == 0 && (ms.q_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(queries) & 15) == 0) {
                const f4* row = reinterpret_cast<const f4*>(Qb + (int64_t)i * dim);
                for (int c = lane; c < (dim >> 2); c += 64) {
                    const f4 a = row[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s2 = fmaf(a[u], a[u], s2);
                }
            } else {
                for (int c = lane; c < dim; c += 64) { const float v = Qb[(int64_t)i * dim + c]; s2 = fmaf(v, v, s2); }
            }
            w_norms += sqrtf(wave_sum(s2));
        }
        __syncthreads();  // (part[] was read above by every thread)
        if (lane == 0) part[wv] = w_norms;
        __syncthreads();
        if (threadIdx.x == 0) hb.m_out[b] = ms.m_abs * ((part[0] + part[1]) + (part[2] + part[3]));
    }
}

// (2) every workgroup finds P = the k-th largest score among the query's G maxima -- ONE wave, the keys in registers (eight per lane), a radix
// select over the 32 score bits by ballots: no barrier, ~1 us (a bitonic network over 512 keys in LDS cost 13 us of barriers in every workgroup)
// -- sets thr[b] = P - 2 m_b and collects its share of the rows that reach it: ids / norms / cnt / flag as collect_above_kernel leaves them, and
// (E != nullptr) the rows themselves into G.  Fewer than k usable maxima (NaN / -inf everywhere): *flag, nothing collected -- the guarded
// full-precision pass answers.
// KV: group maxima per lane of the selecting wave -- 8 (G <= 512: k <= 170) or 32 (G <= 2048, a maximum per wave of 512-score groups: k <= 512, the
// num_hits of the reference's own callers -- 160 to 256, `_search.py:66-67` -- included)
template <int KV>
__global__ __launch_bounds__(256) void pivot_collect_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, const uint64_t* __restrict__ bmax,
                                                             int G, int32_t k, const float* __restrict__ m, float* __restrict__ thr,
                                                             const float* __restrict__ row_norm, int32_t cap, int32_t* __restrict__ ids,
                                                             float* __restrict__ norms, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag,
                                                             const float* __restrict__ E, int dim, float* __restrict__ Gout, int64_t aux_ld, int l2) {
    __shared__ int32_t l_ids[1024];
    __shared__ uint32_t l_n, l_base, pivot_sh;
    const int b = blockIdx.y;
    if (row_norm) row_norm += (int64_t)b * aux_ld;  // (aux_ld != 0: the "norms" are a per-query array -- the scores themselves, launch_topk_pivot)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t kv[KV];
#pragma unroll
        for (int j = 0; j < KV; ++j) kv[j] = lane + 64 * j < G ? (uint32_t)(bmax[(int64_t)b * G + lane + 64 * j] >> 32) : 0u;
        uint32_t need = (uint32_t)k, prefix = 0u;
        for (int bit = 31; bit >= 0; --bit) {  // the keys that agree with `prefix` above `bit` are alive; how many of them have the bit set?
            const uint32_t want = (prefix >> bit) | 1u;
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < KV; ++j) c += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64((kv[j] >> bit) == want));
            if (c >= need) prefix |= 1u << bit;
            else need -= c;
        }
        if (lane == 0) pivot_sh = prefix;  // the k-th largest score key, multiplicity counted (0: fewer than k maxima)
    }
    __syncthreads();
    const uint32_t pivot = pivot_sh;
    if (pivot == 0u) {  // (workgroup-uniform)
        if (blockIdx.x == 0 && threadIdx.x == 0) { thr[b] = INFINITY; atomicOr(flag, 1u); }
        return;
    }
    const float t = lower_threshold(key_score(pivot), m[b], l2 != 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        thr[b] = t;
        if (!(t > -INFINITY)) atomicOr(flag, 1u);
    }
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
    collect_rows(scores + (int64_t)b * ld, n, vec, t, row_norm, cap, ids + (int64_t)b * cap, norms ? norms + (int64_t)b * cap : nullptr, cnt + b, flag,
                 false, 0u, -1, l_ids, &l_n, &l_base, E, dim, Gout ? Gout + (int64_t)b * cap * dim : nullptr);
}

__global__ __launch_bounds__(256) void maxsim_threshold_kernel(const float* __restrict__ topk, int32_t k, const float* __restrict__ Q, int nq,
                                                                int dim, int64_t q_stride, float m_rel, float e_max,
                                                                float* __restrict__ thr, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag,
                                                                const float* __restrict__ q_unscale, float e_norm_max, float* __restrict__ m_out) {
    // |approx - exact| of a chunk's MaxSim score <= sum_i max_j (|q_i| |e_lo,j| + slack |q_i| |e_j|) <= (m_rel * e_max) * sum_i |q_i|
    // with m_rel * e_max = max_j |e_lo,j| + 2^-12 max_j |e_j| handed over by the caller
    // One-product pass (q_unscale != nullptr): the pass multiplied q_hi = fp16(q * q_scale) only, so every pair is also off by
    // q_lo . e with q_lo = q - q_hi / q_scale: |.| <= |q_lo,i| max|e|, and the e_lo term meets |q_hi,i| <= |q_i| + |q_lo,i| -- together
    // e_norm_max * sum_i |q_lo,i| with e_norm_max = max|e| + max|e_lo|.  x - fp16(x) is exact in fp32; the sums are nudged up.
    __shared__ float part[4], part_lo[4];
    const int b = blockIdx.x;
    const float* Qb = Q + (int64_t)b * q_stride;
    const float inv_scale = q_unscale ? q_unscale[2 * b] : 1.0f, q_scale = 1.0f / inv_scale;  // powers of two
    // a wave per query vector (i = wave, wave + 4, ...): no workgroup barrier inside the loop (it had two per vector: 0.05 ms per 128-query step)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float w_norms = 0.f, w_lo = 0.f;
    // (16-byte loads where the layout allows, all of a vector's loads in flight together: with 4-byte loads this kernel took 44 us of every
    // 128-query step -- 128 workgroups waiting for 128 dependent round trips each)
    const bool vec = (dim & 3) == 0 && (q_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0;
    for (int i = wv; i < nq; i += 4) {
        float ss = 0.f, sl = 0.f;
        auto add = [&](float v) {
            ss = fmaf(v, v, ss);
            const float x = v * q_scale;
            const float lo = x - (float)(_Float16)x;
            sl = fmaf(lo, lo, sl);
        };
        if (vec) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4* row = reinterpret_cast<const f4*>(Qb + (int64_t)i * dim);
            const int n4 = dim >> 2;
            int c = lane;
            for (; c + 192 < n4; c += 256) {  // four loads per trip
                const f4 a = row[c], b4 = row[c + 64], c4 = row[c + 128], d4 = row[c + 192];
#pragma unroll
                for (int u = 0; u < 4; ++u) add(a[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) add(b4[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) add(c4[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) add(d4[u]);
            }
            for (; c < n4; c += 64) {
                const f4 a = row[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) add(a[u]);
            }
        } else {
            for (int c = lane; c < dim; c += 64) add(Qb[(int64_t)i * dim + c]);
        }
        w_norms += sqrtf(wave_sum(ss));
        w_lo += sqrtf(wave_sum(sl)) * inv_scale;
    }
    if (lane == 0) { part[wv] = w_norms; part_lo[wv] = w_lo; }
    __syncthreads();
    const float sum_norms = (part[0] + part[1]) + (part[2] + part[3]), sum_lo = (part_lo[0] + part_lo[1]) + (part_lo[2] + part_lo[3]);
    if (threadIdx.x != 0) return;
    float m = m_rel * e_max * sum_norms;
    if (q_unscale) m += e_norm_max * sum_lo * 1.00001f;
    const float t = topk[(int64_t)b * k + (k - 1)] - 2.0f * m;
    thr[b] = t;
    cnt[b] = 0u;
    if (m_out) m_out[b] = m;
    if (!(t > -INFINITY)) atomicOr(flag, 1u);  // NaN or -inf: fewer than k scorable chunks
}

// The second, tighter threshold of a MaxSim batch.  exact[b][0 .. k) = the EXACT scores of query b's approximate top-k chunks top_i[b][.]
// (maxsim_pairs_kernel).  L = their minimum is the k-th best exact score of a k-subset, hence <= the k-th best exact score overall; a
// chunk of the exact top-k has exact >= L and |approx - exact| <= m_b, so its approximate score is >= L - m_b =: thr[b] -- against
// (k-th best approximate) - 2 m_b of the first threshold, and never below it (every member of the approximate top-k has exact >=
// approx - m >= k-th approximate - m).  The approximate top-k itself passes (approx >= exact - m >= L - m): it becomes the head of the
// candidate list -- ids[b][0 .. k), es[b][0 .. k), cnt[b] = k -- and collect_above_kernel appends only what ranks below it.
// Fewer than k scorable chunks (an id < 0, a NaN): *flag, nothing collected (thr = +inf): the guarded full-precision path answers.
// qsum != nullptr (a batch whose query image carries the sums, maxsim_gemm.hip: query_planes_kernel): m_b is computed HERE -- m_abs sum_i |q_i|
// (+ e_norm_max sum_i |q_lo,i| for the one-product pass, the statement of maxsim_threshold_kernel) -- and left in m[b]; no threshold kernel
// ran before.  Always: the slots k .. cap - 1 of the list are set to -1 ("no chunk"), so no memset launch precedes the collection.
__global__ __launch_bounds__(256) void exact_threshold_kernel(const float* __restrict__ exact, const int32_t* __restrict__ top_i, int32_t k,
                                                               float* __restrict__ m, int32_t cap, float* __restrict__ thr,
                                                               uint32_t* __restrict__ cnt, int32_t* __restrict__ ids, float* __restrict__ es,
                                                               uint32_t* __restrict__ flag, const float* __restrict__ qsum, float m_abs,
                                                               float e_norm_max, int with_lo, const float* __restrict__ top_s) {
    __shared__ float part[4];
    __shared__ int bad_sh;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) bad_sh = 0;
    __syncthreads();
    float mn = INFINITY;
    bool bad = false;
    for (int j = threadIdx.x; j < k; j += 256) {
        const float v = exact[(int64_t)b * k + j];
        const int32_t c = top_i[(int64_t)b * k + j];
        bad |= c < 0 || !(v > -INFINITY);
        // (a member of the approximate top-k whose approximate score is -inf is a MASKED chunk -- a tombstone, a metadata filter: fewer than k
        // chunks are eligible, its exact score must not count; the guarded pass, which masks its scores, answers)
        if (top_s) bad |= !(top_s[(int64_t)b * k + j] > -INFINITY);
        mn = fminf(mn, v);
        ids[(int64_t)b * cap + j] = c;
        es[(int64_t)b * cap + j] = v;
    }
    for (int j = k + threadIdx.x; j < cap; j += 256) ids[(int64_t)b * cap + j] = -1;  // (the list's unused slots: "no chunk" -- no memset launch)
    if (bad) bad_sh = 1;  // (benign race: every writer stores 1)
    mn = -wave_max(-mn);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float L = fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
    float mb;
    if (qsum) {
        mb = m_abs * qsum[2 * b];
        if (with_lo) mb += e_norm_max * qsum[2 * b + 1] * 1.00001f;
        m[b] = mb;
    } else {
        mb = m[b];
    }
    float t = L - mb * 1.00001f;
    t -= fabsf(t) * 0x1p-22f;  // (the subtraction's own rounding)
    const bool unusable = bad_sh != 0 || k > cap || !(t > -INFINITY) || !(mb >= 0.f);
    thr[b] = unusable ? INFINITY : t;
    cnt[b] = unusable ? 0u : (uint32_t)k;
    if (unusable) atomicOr(flag, 1u);
}

// Batched flavour (api.hip: search_rows_fused_hi, experimental), one block per query: thr[b] = topk[b * k + k - 1] - window[b] with
// window[b] = 2 m_b the width of the error band of query b's approximate similarities:
//   cosine: m = lo_ratio + 2^-12 + (1 + lo_ratio) |q_lo| / |q|        dot: m = (lo_norm + 2^-12 e_norm) |q| + (e_norm + lo_norm) |q_lo| + 2^-22
// (lo_ratio = max |e_lo| / |e|, lo_norm = max |e_lo|, e_norm = max |e| over the rows; the |q_lo| terms only when q_unscale is given,
// i.e. when the pass multiplied the queries' fp16 hi halves only: q_lo = q - fp16(q * scale) / scale with scale = 1 / q_unscale[b],
// the statement of query_rows_planes_kernel).  Zeroes cnt[b] and cnt2[b]; an unusable threshold sets *flag.
__global__ __launch_bounds__(256) void row_threshold_kernel(const float* __restrict__ topk, int32_t k, const float* __restrict__ Q, int dim, int mode,
                                                             const float* __restrict__ q_unscale, float lo_ratio, float lo_norm, float e_norm,
                                                             float* __restrict__ thr, float* __restrict__ window, uint32_t* __restrict__ cnt,
                                                             uint32_t* __restrict__ cnt2, uint32_t* __restrict__ flag, float* __restrict__ thr_copy, float sum_eps) {
    __shared__ float part[4], part_lo[4];
    const int b = blockIdx.x;
    const float* q = Q + (int64_t)b * dim;
    const float inv_scale = q_unscale ? q_unscale[b] : 1.0f, q_scale = 1.0f / inv_scale;  // powers of two
    float ss = 0.f, sl = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float v = q[c];
        ss = fmaf(v, v, ss);
        const float x = v * q_scale;
        const float lo = x - (float)(_Float16)x;
        sl = fmaf(lo, lo, sl);
    }
    ss = wave_sum(ss);
    sl = wave_sum(sl);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = ss; part_lo[threadIdx.x >> 6] = sl; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float qn = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
    const float ql = q_unscale ? sqrtf((part_lo[0] + part_lo[1]) + (part_lo[2] + part_lo[3])) * inv_scale * 1.00001f : 0.f;
    float m;
    if (mode == SCAN_COSINE) m = lo_ratio + sum_eps + (1.0f + lo_ratio) * (ql / qn) * 1.00001f;
    else m = (lo_norm + sum_eps * e_norm) * qn + (e_norm + lo_norm) * ql + 0x1p-22f;
    const float w = 2.0f * m * 1.00001f;
    const float t = topk[(int64_t)b * k + (k - 1)] - w;
    thr[b] = t;
    if (thr_copy) thr_copy[b] = t;
    window[b] = w;
    cnt[b] = 0u;
    cnt2[b] = 0u;
    if (!(t > -INFINITY) || !(w >= 0.f)) atomicOr(flag, 1u);  // NaN or -inf: fewer than k usable sample scores, or a degenerate query
}

// Exact similarities of (query, row) pairs: out[b * cap + p] = metric(q_b . e_row) for row = rows[b * cap + p], p < cnt[b] -- one wave
// per pair, fp32 FMAs in a fixed order (lane l sums elements 4 l .. 4 l + 3 of every 256, then the butterfly), the metric by the
// statements of transform_score.  What the batched half-bytes search ranks its candidates by (a few hundred per query).
__global__ __launch_bounds__(256) void row_dots_kernel(const float* __restrict__ E, int dim, const float* __restrict__ Q, const int32_t* __restrict__ rows,
                                                        const uint32_t* __restrict__ cnt, int32_t cap, int mode, const float* __restrict__ row_norm,
                                                        const float* __restrict__ q_sumsq, float* __restrict__ out) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int n = (int)min(cnt[b], (uint32_t)cap);
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const float* q = Q + (int64_t)b * dim;
    for (int p = wave; p < n; p += n_waves) {  // (wave-uniform)
        const int32_t row = rows[(int64_t)b * cap + p];
        const float* e = E + (int64_t)row * dim;
        float acc = 0.f;
        for (int c = 4 * lane; c < dim; c += 256) {
            const f4 ev = *reinterpret_cast<const f4*>(e + c), qv = *reinterpret_cast<const f4*>(q + c);
            acc = fmaf(ev[0], qv[0], acc);
            acc = fmaf(ev[1], qv[1], acc);
            acc = fmaf(ev[2], qv[2], acc);
            acc = fmaf(ev[3], qv[3], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const float qss = mode == SCAN_COSINE ? q_sumsq[b] : 0.f;
            out[(int64_t)b * cap + p] = transform_score(acc, mode, mode == SCAN_COSINE ? row_norm[row] : 0.f, 0.f, sqrtf(qss), qss);
        }
    }
}

// bits[0] = max |e|, bits[1] = max |e_lo|, bits[2] = max |e_lo| / |e|, bits[3] = min |e| over the rows (float bit patterns, the maxima nudged up by 1e-6;
// non-negative floats order like their bits), where e_lo = e - fp16(e * scale) / scale is what the HI halves (rounded to nearest even,
// as the plane and the image are built) drop -- computed exactly (the scale is a power of two, x - fp16(x) is exact in fp32).
__global__ __launch_bounds__(256) void max_row_norm_kernel(const float* __restrict__ E, int64_t n_rows, int dim, float scale,
                                                            uint32_t* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const float inv = 1.0f / scale;
    float mx = 0.f, mlo = 0.f, mratio = 0.f, mn = INFINITY;
    for (int64_t r = wave0; r < n_rows; r += n_waves) {
        float ss = 0.f, sl = 0.f;
        for (int c = lane; c < dim; c += 64) {
            const float v = E[r * (int64_t)dim + c];
            const float x = v * scale;
            const float hi = (float)(_Float16)x;  // (the rounding the planes were built with)
            const float lo = (x - hi) * inv;
            ss = fmaf(v, v, ss);
            sl = fmaf(lo, lo, sl);
        }
        ss = wave_sum(ss);
        sl = wave_sum(sl);
        const float ne = sqrtf(ss), nl = sqrtf(sl);
        mx = fmaxf(mx, ne);
        mn = fminf(mn, ne);
        mlo = fmaxf(mlo, nl);
        if (ne > 0.f) mratio = fmaxf(mratio, nl / ne);
    }
    if (lane == 0) {
        if (mn < INFINITY) atomicMin(bits + 3, __float_as_uint(mn));  // bits[3] = min |e| (the caller starts it at its current minimum)
        if (mx > 0.f) atomicMax(bits + 0, __float_as_uint(mx * 1.000001f));
        if (mlo > 0.f) atomicMax(bits + 1, __float_as_uint(mlo * 1.000001f));
        if (mratio > 0.f) atomicMax(bits + 2, __float_as_uint(mratio * 1.000001f));
    }
}

// The same maximum over an fp16-STORED corpus (bits[0] only: the stored halves are the corpus, nothing is dropped)
__global__ __launch_bounds__(256) void max_row_norm16_kernel(const uint16_t* __restrict__ E, int64_t n_rows, int dim, uint32_t* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    float mx = 0.f;
    for (int64_t r = wave0; r < n_rows; r += n_waves) {
        float ss = 0.f;
        for (int c = lane; c < dim; c += 64) {
            const float v = (float)reinterpret_cast<const _Float16*>(E)[r * (int64_t)dim + c];
            ss = fmaf(v, v, ss);
        }
        mx = fmaxf(mx, sqrtf(wave_sum(ss)));
    }
    if (lane == 0 && mx > 0.f) atomicMax(bits + 0, __float_as_uint(mx * 1.000001f));
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

// ---- a MaxSim batch over a SHARDED corpus: one threshold for all shards (api.hip: rl_maxsim_batch_begin / _finish) -------------------------
// out[b][0 .. k) = this shard's k best approximate scores of query b (descending), out[b][k] = its bound m_b (neg2m = -2 m_b)
__global__ __launch_bounds__(256) void pack_approx_kernel(const float* __restrict__ top, const float* __restrict__ neg2m, int32_t k, int64_t count,
                                                           float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int64_t b = i / (k + 1);
    const int32_t j = (int32_t)(i - b * (k + 1));
    out[i] = j < k ? top[b * k + j] : -0.5f * neg2m[b];
}

// One block per query.  lists [world][B][k + 1]: every shard's k best approximate scores (descending) and its bound.  A = the k-th best
// approximate score over ALL shards (the global top-k by approximate score lies inside the union of the shards' top-k), m* = the
// largest bound: at least k chunks have an exact score >= A - m*, so a chunk of the exact global top-k on this shard has an exact score
// >= A - m* and an approximate one >= A - m* - m_rank.  thr[b] = that; cnt[b] = 0; an unusable threshold (fewer than k scorable chunks
// over all shards, NaN) sets *flag: this shard's guarded full-precision path then returns ITS exact top-k, which the merge accepts as well.
__global__ __launch_bounds__(256) void global_threshold_kernel(const float* __restrict__ lists, int32_t world, int32_t B, int32_t k, int32_t rank,
                                                                float* __restrict__ thr, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag) {
    __shared__ float part[4];
    const int b = blockIdx.x;
    const int64_t stride = (int64_t)B * (k + 1);
    const float* mine = lists + (int64_t)b * (k + 1);
    float best = -INFINITY;  // the largest value with at least k values >= it
    for (int i = threadIdx.x; i < world * k; i += 256) {
        const float v = mine[(int64_t)(i / k) * stride + (i % k)];
        if (!(v > -INFINITY)) continue;  // (-inf padding and NaN never are the k-th best of k scorable chunks)
        int32_t c = 0;
        for (int r = 0; r < world && c < k; ++r) {  // how many of shard r's values are >= v: its list is sorted, descending
            const float* l = mine + (int64_t)r * stride;
            int32_t lo = 0, hi = k;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (l[mid] >= v) lo = mid + 1; else hi = mid;
            }
            c += lo;
        }
        if (c >= k) best = fmaxf(best, v);
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float A = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    float m_max = 0.f;
    for (int r = 0; r < world; ++r) m_max = fmaxf(m_max, mine[(int64_t)r * stride + k]);
    const float t = A - (m_max + mine[(int64_t)rank * stride + k]) * 1.00001f;
    thr[b] = t;
    cnt[b] = 0u;
    if (!(t > -INFINITY)) atomicOr(flag, 1u);
}

int launch_pack_approx(const float* top, const float* neg2m, int32_t B, int32_t k, float* out, hipStream_t s) {
    const int64_t count = (int64_t)B * (k + 1);
    if (count <= 0) return RL_OK;
    hipLaunchKernelGGL(pack_approx_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, top, neg2m, k, count, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_global_threshold(const float* lists, int32_t world, int32_t B, int32_t k, int32_t rank, float* thr, uint32_t* cnt, uint32_t* flag,
                            hipStream_t s) {
    if (B <= 0) return RL_OK;
    hipLaunchKernelGGL(global_threshold_kernel, dim3(B), dim3(256), 0, s, lists, world, B, k, rank, thr, cnt, flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_approx_threshold(const float* topk, int32_t nb, int32_t k, const float* queries, int32_t dim, int mode, float m_rel,
                            float e_norm_bound, float* thr, uint32_t* cnt, uint32_t* flag, hipStream_t s, float e_max) {
    hipLaunchKernelGGL(approx_threshold_kernel, dim3(1), dim3(256), 0, s, topk, nb, k, queries, (int)dim, mode, m_rel, e_norm_bound, thr, cnt,
                       flag, e_max);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// The pivot route (see transform_bmax_kernel): candidates of a bound-filtered search without ranking the approximate scores.  bmax:
// pivot_scratch_words(nb) 8-byte words.  RL_ERR_UNSUPPORTED where the route does not pay (fewer than 3 k group maxima, k > 512).
// maxsim != nullptr: the scores are a MaxSim query's chunk scores (mode raw; `queries` = the queries' vectors, maxsim->q_stride floats apart):
// bound m_b = m_abs * sum_i |q_i|, the lists pre-filled with -1 (maxsim_pairs_kernel walks every slot).
constexpr int PIVOT_MAX_K = 512, PIVOT_MAX_G = 2048;
size_t pivot_scratch_words(int32_t nb) { return (size_t)nb * PIVOT_MAX_G; }
// The groups whose maxima the pivot is taken from: a maximum per workgroup of 2048 scores (up to 512 of them); a query with fewer scores than 3 k such
// groups gets one per WAVE (up to 4 x 128); k > 170 one per wave of up to 512 workgroups (G <= 2048).  false: fewer than 3 k groups -- no pivot.
static bool pivot_groups(int64_t n, int32_t k, int* bx_out, int* G_out, int* per_wave_out) {
    if (n <= 0 || k < 1 || k > PIVOT_MAX_K) return false;
    int bx = (int)std::min<int64_t>((n + 2047) / 2048, 512), G = bx, per_wave = 0;
    if (G < 3 * k) {
        bx = (int)std::min<int64_t>((n + 1023) / 1024, 128);
        G = 4 * bx;
        per_wave = 1;
    }
    if (G < 3 * k) {
        bx = (int)std::min<int64_t>((n + 1023) / 1024, 512);
        G = 4 * bx;
    }
    if (G < 3 * k) return false;
    *bx_out = bx; *G_out = G; *per_wave_out = per_wave;
    return true;
}
// does launch_pivot_route take n scores per query at this k?  (its own test, for callers that have no other way to find candidates)
bool pivot_route_takes(int64_t n, int32_t k) {
    int bx, G, pw;
    return pivot_groups(n, k, &bx, &G, &pw);
}
int launch_pivot_route(float* scores, int32_t nb, int64_t n, int64_t ld, int32_t k, const float* row_norm, const float* row_sumsq, const float* queries,
                       int32_t dim, int mode, float pre_scale, uint64_t* bmax, uint32_t* zero_words, int n_zero, const HiBound& bound, float* thr,
                       int32_t cap, int32_t* ids, float* norms, uint32_t* cnt, uint32_t* flag, hipStream_t s, const float* E, float* gather_out,
                       bool* gathered, const PivotMaxSim* maxsim, const float* aux_src, int64_t aux_ld) {
    if (gathered) *gathered = false;
    if (n <= 0 || nb <= 0 || k < 1 || k > PIVOT_MAX_K || !bound.m_out) return RL_ERR_UNSUPPORTED;
    if (n_zero < 0 || n_zero > 256 || (n_zero > 0 && !zero_words)) return RL_ERR_INVALID;
    PivotMaxSim ms = maxsim ? *maxsim : PivotMaxSim{};
    if (maxsim && ms.nq > 0 && mode != SCAN_RAW_DOT) return RL_ERR_INVALID;
    int bx = 0, G = 0, per_wave = 0;
    if (!pivot_groups(n, k, &bx, &G, &per_wave)) return RL_ERR_UNSUPPORTED;
    ms.per_wave = per_wave;
    if (ms.fill_ids) ms.fill_n = (int64_t)nb * cap;
    hipLaunchKernelGGL(transform_bmax_kernel, dim3(bx, nb), dim3(256), 0, s, scores, n, ld, row_norm, row_sumsq, queries, (int)dim, mode, pre_scale, bmax,
                       zero_words, n_zero, bound, ms);
    const int cx = (int)std::max<int64_t>(1, std::min<int64_t>((n + 4095) / 4096, 512));
    // (the candidates' rows gathered by the collecting workgroups themselves where the layout allows 16-byte copies)
    const bool fuse = E && gather_out && gathered && (dim & 3) == 0 && ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(gather_out)) & 15) == 0;
    // (aux_src: what goes into `norms` next to every collected id -- the rows' norms for a cosine search, or any other per-row / per-query array)
    const float* aux = aux_src ? aux_src : (mode == SCAN_COSINE ? row_norm : nullptr);
    if (G <= 512)
        hipLaunchKernelGGL(pivot_collect_kernel<8>, dim3(cx, nb), dim3(256), 0, s, scores, n, ld, bmax, G, k, bound.m_out, thr, aux, cap, ids, norms, cnt, flag,
                           fuse ? E : nullptr, (int)dim, fuse ? gather_out : nullptr, aux_src ? aux_ld : (int64_t)0, mode == SCAN_L2 ? 1 : 0);
    else
        hipLaunchKernelGGL(pivot_collect_kernel<32>, dim3(cx, nb), dim3(256), 0, s, scores, n, ld, bmax, G, k, bound.m_out, thr, aux, cap, ids, norms, cnt, flag,
                           fuse ? E : nullptr, (int)dim, fuse ? gather_out : nullptr, aux_src ? aux_ld : (int64_t)0, mode == SCAN_L2 ? 1 : 0);
    if (fuse) *gathered = true;
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_collect_above(const float* scores, int32_t nb, int64_t n, int64_t ld, const float* thr, const float* row_norm, int32_t cap,
                         int32_t* ids, float* norms, uint32_t* cnt, uint32_t* flag, hipStream_t s, const float* top_s, const int32_t* top_i,
                         int32_t k) {
    if (n <= 0 || nb <= 0) return RL_OK;
    if ((top_s != nullptr) != (top_i != nullptr) || (top_s && k < 1)) return RL_ERR_INVALID;
    // (a batch of many queries: 16 k scores per workgroup instead of 4 k -- a quarter of the workgroups to dispatch, eight loads per lane in a loop
    // that keeps two in flight; one query keeps the fine grid, which is what fills the chip there)
    const int64_t per_wg = nb >= 32 ? 16384 : 4096;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((n + per_wg - 1) / per_wg, 512));
    hipLaunchKernelGGL(collect_above_kernel, dim3(bx, nb), dim3(256), 0, s, scores, n, ld, thr, row_norm, cap, ids, norms, cnt, flag, top_s, top_i, k);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- fp16 queries over an fp16-STORED corpus (rl_maxsim_topk_batch_f16): the one-product pass IS the score --------------------------
// RAGLite's stored embeddings AND its query embeddings are fp16 values (src/raglite/_embed.py:140,164; the adapter's result is cast back,
// src/raglite/_search.py:62).  The product of two fp16 values is exact in fp32, the index stores e itself (no e_lo), and a query whose
// elements survive its own power-of-two scaling has no q_lo: what maxsim_pp_kernel accumulates is then the fp32-accumulated dot product
// itself -- no bound, no candidate list, no re-scoring.  This kernel certifies that per query and hands the pass's top-k out as the result:
// a query with a lost bit (elements spread over more than fp16's exponent range around the scaled maximum), or without k scorable chunks,
// raises the device flag and the guarded full-precision passes answer the batch instead.
// ALIGNED: src is 8-byte aligned (one 8-byte load per four elements); else four 2-byte loads (a contiguous slice of a caller's fp16
// array may start at any element: the values are what matters, not where they lie)
template <bool ALIGNED>
__global__ __launch_bounds__(256) void widen_f16_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t count) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < count) {
        _Float16 h[4];
        if constexpr (ALIGNED) {
            const uint2 v = *reinterpret_cast<const uint2*>(src + i);  // (i % 4 == 0 and the buffer is 8-byte aligned)
            __builtin_memcpy(h, &v, 8);
        } else {
            __builtin_memcpy(h, src + i, 8);  // (2-byte aligned only: the compiler emits element loads)
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f4*>(dst + i) = (f4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        for (int64_t j = i; j < count; ++j) {
            _Float16 h;
            __builtin_memcpy(&h, src + j, 2);
            dst[j] = (float)h;
        }
    }
}

__global__ __launch_bounds__(256) void f16_exact_finish_kernel(const float* __restrict__ Q, int nq, int dim, int64_t q_stride,
                                                                const float* __restrict__ q_unscale, const float* __restrict__ top_s,
                                                                const int32_t* __restrict__ top_i, int32_t k, float* __restrict__ out_s,
                                                                int32_t* __restrict__ out_i, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flag) {
    const int b = blockIdx.x;
    const float* Qb = Q + (int64_t)b * q_stride;
    const float q_scale = 1.0f / q_unscale[2 * b];  // a power of two (query_planes_kernel)
    bool lost = false;
    for (int64_t c = threadIdx.x; c < (int64_t)nq * dim; c += 256) {
        const float x = Qb[c] * q_scale;
        lost |= (float)(_Float16)x != x;  // (also true for NaN / inf elements)
    }
    if (__syncthreads_or(lost ? 1 : 0) || !(top_s[(int64_t)b * k + (k - 1)] > -INFINITY)) {
        if (threadIdx.x == 0) atomicOr(flag, 1u);
    }
    for (int j = threadIdx.x; j < k; j += 256) {
        out_s[(int64_t)b * k + j] = top_s[(int64_t)b * k + j];
        out_i[(int64_t)b * k + j] = top_i[(int64_t)b * k + j];
    }
    if (threadIdx.x == 0) cnt[b] = 0u;  // (rl_index_filter_stats: no candidate is re-scored on this route)
}

__global__ __launch_bounds__(256) void scale_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, float factor, int64_t count,
                                                         uint32_t* __restrict__ zero_words, int n_zero) {
    if (blockIdx.x == 0 && (int)threadIdx.x < n_zero) zero_words[threadIdx.x] = 0u;  // (a small hipMemsetAsync is THREE fill launches)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i] * factor;
}

int launch_widen_f16(const uint16_t* src, float* dst, int64_t count, hipStream_t s) {
    if (count <= 0) return RL_OK;
    if ((reinterpret_cast<uintptr_t>(src) & 1) || (reinterpret_cast<uintptr_t>(dst) & 15)) return RL_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((count + 1023) / 1024));
    if (reinterpret_cast<uintptr_t>(src) & 7) hipLaunchKernelGGL(widen_f16_kernel<false>, grid, dim3(256), 0, s, src, dst, count);
    else hipLaunchKernelGGL(widen_f16_kernel<true>, grid, dim3(256), 0, s, src, dst, count);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_f16_exact_finish(const float* Q, int32_t nq, int32_t dim, int64_t q_stride, const float* q_unscale, const float* top_s,
                            const int32_t* top_i, int32_t n_queries, int32_t k, float* out_s, int32_t* out_i, uint32_t* cnt, uint32_t* flag,
                            hipStream_t s) {
    if (n_queries <= 0) return RL_OK;
    hipLaunchKernelGGL(f16_exact_finish_kernel, dim3(n_queries), dim3(256), 0, s, Q, (int)nq, (int)dim, q_stride, q_unscale, top_s, top_i, k, out_s,
                       out_i, cnt, flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_scale_f32(const float* src, float* dst, float factor, int64_t count, hipStream_t s, uint32_t* zero_words, int n_zero) {
    if (count <= 0) return RL_OK;
    if (n_zero < 0 || n_zero > 256 || (n_zero > 0 && !zero_words)) return RL_ERR_INVALID;
    hipLaunchKernelGGL(scale_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src, dst, factor, count, zero_words, n_zero);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_maxsim_threshold(const float* topk, int32_t n_queries, int32_t k, const float* Q, int32_t nq, int32_t dim, int64_t q_stride,
                            float m_rel, float e_max, float* thr, uint32_t* cnt, uint32_t* flag, hipStream_t s, const float* q_unscale,
                            float e_norm_max, float* m_out) {
    if (n_queries <= 0) return RL_OK;
    hipLaunchKernelGGL(maxsim_threshold_kernel, dim3(n_queries), dim3(256), 0, s, topk, k, Q, (int)nq, (int)dim, q_stride, m_rel, e_max, thr, cnt,
                       flag, q_unscale, e_norm_max, m_out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_exact_threshold(const float* exact, const int32_t* top_i, int32_t n_queries, int32_t k, float* m, int32_t cap, float* thr,
                           uint32_t* cnt, int32_t* ids, float* es, uint32_t* flag, hipStream_t s, const float* qsum, float m_abs, float e_norm_max,
                           bool with_lo, const float* top_s) {
    if (n_queries <= 0) return RL_OK;
    hipLaunchKernelGGL(exact_threshold_kernel, dim3(n_queries), dim3(256), 0, s, exact, top_i, k, m, cap, thr, cnt, ids, es, flag, qsum, m_abs,
                       e_norm_max, with_lo ? 1 : 0, top_s);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_row_threshold(const float* topk, int32_t nb, int32_t k, const float* Q, int32_t dim, int mode, const float* q_unscale, float lo_ratio,
                         float lo_norm, float e_norm, float* thr, float* window, uint32_t* cnt, uint32_t* cnt2, uint32_t* flag, hipStream_t s,
                         float* thr_copy, float sum_eps) {
    if (nb <= 0) return RL_OK;
    hipLaunchKernelGGL(row_threshold_kernel, dim3(nb), dim3(256), 0, s, topk, k, Q, (int)dim, mode, q_unscale, lo_ratio, lo_norm, e_norm, thr, window,
                       cnt, cnt2, flag, thr_copy, sum_eps);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_row_dots(const float* E, int32_t dim, const float* Q, int32_t nb, const int32_t* rows, const uint32_t* cnt, int32_t cap, int mode,
                    const float* row_norm, const float* q_sumsq, float* out, hipStream_t s) {
    if (nb <= 0 || cap <= 0) return RL_OK;
    if (dim % 4 || (reinterpret_cast<uintptr_t>(E) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    if (mode != SCAN_COSINE && mode != SCAN_DOT) return RL_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(row_dots_kernel, dim3(8, nb), dim3(256), 0, s, E, (int)dim, Q, rows, cnt, cap, mode, row_norm, q_sumsq, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_max_row_norm(const float* E, int64_t n_rows, int32_t dim, float scale, uint32_t* bits, hipStream_t s) {
    if (n_rows <= 0) return RL_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_rows + 3) / 4, 256 * 8));
    hipLaunchKernelGGL(max_row_norm_kernel, dim3(blocks), dim3(256), 0, s, E, n_rows, (int)dim, scale, bits);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_max_row_norm16(const uint16_t* E, int64_t n_rows, int32_t dim, uint32_t* bits, hipStream_t s) {
    if (n_rows <= 0) return RL_OK;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_rows + 3) / 4, 256 * 8));
    hipLaunchKernelGGL(max_row_norm16_kernel, dim3(blocks), dim3(256), 0, s, E, n_rows, (int)dim, bits);
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
