// a7 / a8 / section 8e: exact top-k selection, per-chunk max grouping, shard merge.
//
// a7 replaces `ORDER BY dist LIMIT num_hits` (src/raglite/_search.py:75-79) with an EXACT selection
// (the reference's HNSW index is approximate).  Order: (score descending, id ascending) -- SQL leaves
// ties unspecified; the oracle and this file fix them to the lowest id.  NaN scores rank last.
//
// Every element is mapped to a unique 64-bit key (orderable score bits << 32 | ~id); top-k = the k
// largest keys.  Three launches per batch of queries, no host round trip, no memset (the histogram rows are zero between
// selections: the final kernel zeroes the row it has just read):
//   1. hist    : 2048-bin histogram of the key's top 11 bits (LDS-privatised, four lane-interleaved copies, nonzero bins
//                flushed).  For raw dots that still need their metric transform (small batches over the MFMA stream)
//                transform_hist_kernel does both in one launch.
//   2. filter  : every block locates the threshold bin from the histogram; elements in higher bins go
//                straight to the `sel` list, elements in the threshold bin to the `cand` list.
//   3. final   : one block per query bitonic-sorts the (few hundred) candidates, takes what is still
//                needed, sorts the k survivors and writes (score, id).  If the threshold bin overflows
//                CAND_CAP (massive ties, e.g. the reference's all-ones test corpus,
//                tests/test_split_chunks.py:28) the block refines the threshold itself with two more
//                radix passes over the scores and an index-ordered tie pass -- slow but exact.
// Traffic: scores are read twice (4*N B each; 0.2 % of the 4*N*dim B scan that produced them).
#include "common.h"

namespace rl {

constexpr int CNT_SEL = HIST_BINS + 0;   // counters live right after the bins
constexpr int CNT_CAND = HIST_BINS + 1;
constexpr int HIST_STRIDE = HIST_BINS + 8;
constexpr int RANK_MAX = 1024;        // up to this many keys are ordered by counting instead of a bitonic network

// ---- block-wide helpers (256 or 1024 threads) -----------------------------------------------------

// Inclusive prefix sum over the block; `scratch` holds >= blockDim/64 + 1 uint32.  Returns the inclusive
// value for the caller and the block total through `total`.
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* scratch, uint32_t& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) scratch[w] = x;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < w; ++i) base += scratch[i];
    uint32_t t = 0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    total = t;
    return x + base;
}

// Given `hist[0..nbins)` in LDS (nbins <= 8 * blockDim), find the bin b* such that the number of
// elements in bins > b* is < need <= that number + hist[b*].  Results in sh_out[0] = b*, sh_out[1] =
// count above.  Every thread must call; ends with a barrier.  If the histogram holds fewer than `need`
// elements, b* = 0 and above = total - hist[0].
__device__ __forceinline__ void find_threshold_bin(const uint32_t* hist, int nbins, uint32_t need, uint32_t* scratch,
                                                    uint32_t* sh_out) {
    const int per = (nbins + (int)blockDim.x - 1) / (int)blockDim.x;  // bins per thread, walked from the top
    const int hi = nbins - 1 - (int)threadIdx.x * per;
    uint32_t s = 0;
    for (int j = 0; j < per; ++j) {
        const int b = hi - j;
        if (b >= 0) s += hist[b];
    }
    uint32_t total;
    const uint32_t incl = block_inclusive_scan(s, scratch, total);
    const uint32_t excl = incl - s;
    if (threadIdx.x == 0) { sh_out[0] = 0; sh_out[1] = total - hist[0]; }
    __syncthreads();
    if (excl < need && need <= incl) {
        uint32_t cum = excl;
        for (int j = 0; j < per; ++j) {
            const int b = hi - j;
            if (b < 0) break;
            const uint32_t c = hist[b];
            if (cum + c >= need) { sh_out[0] = (uint32_t)b; sh_out[1] = cum; break; }
            cum += c;
        }
    }
    __syncthreads();
}

// In-LDS bitonic sort, descending, n a power of two.
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* a, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = a[i], y = a[ixj];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- 1. histogram -----------------------------------------------------------------------------------
// Similarities of one query crowd into a few dozen bins (cosines around 0: ~30 bins hold 95 % of a corpus), and LDS atomics of
// one instruction that hit the same address serialise: round 1 measured 85 % of this kernel's LDS cycles as conflicts
// (profiles/r01_m_pmc.txt).  Four copies of the histogram, copy = lane & 3, 16 banks apart: a quarter of the collisions.
constexpr int HIST_COPIES = 4;
constexpr int HIST_COPY_STRIDE = HIST_BINS + 16;
__device__ __forceinline__ void hist_zero(uint32_t* h) {
    for (int i = threadIdx.x; i < HIST_COPIES * HIST_COPY_STRIDE; i += blockDim.x) h[i] = 0;
}
__device__ __forceinline__ void hist_add(uint32_t* h, float v) {
    atomicAdd(&h[(threadIdx.x & (HIST_COPIES - 1)) * HIST_COPY_STRIDE + (score_key(v) >> 21)], 1u);
}
__device__ __forceinline__ void hist_flush(const uint32_t* h, uint32_t* g) {
    for (int i = threadIdx.x; i < HIST_BINS; i += blockDim.x) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < HIST_COPIES; ++j) c += h[j * HIST_COPY_STRIDE + i];
        if (c) atomicAdd(&g[i], c);
    }
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ scores, int64_t n, int64_t ld,
                                                         uint32_t* __restrict__ ws_hist, const uint32_t* __restrict__ run_if) {
    __shared__ uint32_t h[HIST_COPIES * HIST_COPY_STRIDE];
    if (run_if && *run_if == 0u) return;  // guarded fallback of the fused top-k: not needed
    const int q = blockIdx.y;
    hist_zero(h);
    __syncthreads();
    const float* s = scores + (int64_t)q * ld;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {  // 16-B loads: 4 elements per lane in flight
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* s4 = reinterpret_cast<const f4*>(s);
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f4 v = s4[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) hist_add(h, v[u]);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) hist_add(h, s[(n4 << 2) + threadIdx.x]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) hist_add(h, s[i]);
    }
    __syncthreads();
    hist_flush(h, ws_hist + (int64_t)q * HIST_STRIDE);
}

// Raw dots -> similarities in place (the statements of scan.hip:transform_kernel) AND their histogram: one launch and one
// pass over the scores instead of transform + memset + hist (B = 1 over 1 M rows: 5.3 + 4.7 + 6.5 us of kernels and two
// launch gaps, profiles/r02_m_cfg2_kernel_stats.csv).
__global__ __launch_bounds__(256) void transform_hist_kernel(float* __restrict__ scores, int64_t n, int64_t ld,
                                                              const float* __restrict__ row_norm,
                                                              const float* __restrict__ row_sumsq,
                                                              const float* __restrict__ queries, int dim, int mode,
                                                              uint32_t* __restrict__ ws_hist, float pre_scale,
                                                              const uint32_t* __restrict__ run_if, uint32_t* __restrict__ zero_words,
                                                              int n_zero, HiBound hb) {
    __shared__ uint32_t h[HIST_COPIES * HIST_COPY_STRIDE];
    __shared__ float part[4];
    if (run_if && *run_if == 0u) return;
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && b == 0 && (int)threadIdx.x < n_zero) zero_words[threadIdx.x] = 0u;
    typedef float f4 __attribute__((ext_vector_type(4)));
    float* const sb = scores + (int64_t)b * ld;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;  // 16-B accesses (row_norm / row_sumsq are hipMalloc'd)
    const int64_t n4 = vec ? n >> 2 : 0;
    // The kernel is latency-bound (4 MB at B = 1): two 16-B groups per lane and trip, and the first trip's loads are issued
    // BEFORE the query-norm reduction and the histogram reset so that their latency hides behind both.
    struct Trip { f4 d0, d1, a0, a1; int64_t i0, i1; bool has0, has1; };
    auto fetch = [&](int64_t i) {
        Trip t;
        t.i0 = i; t.i1 = i + stride;
        t.has0 = t.i0 < n4; t.has1 = t.i1 < n4;
        const int64_t j0 = t.has0 ? t.i0 : 0, j1 = t.has1 ? t.i1 : j0;
        t.d0 = t.d1 = t.a0 = t.a1 = (f4){0.f, 0.f, 0.f, 0.f};
        if (n4 > 0) {
            t.d0 = reinterpret_cast<const f4*>(sb)[j0];
            t.d1 = reinterpret_cast<const f4*>(sb)[j1];
            const float* aux = mode == SCAN_COSINE ? row_norm : mode == SCAN_L2 ? row_sumsq : nullptr;
            if (aux) { t.a0 = reinterpret_cast<const f4*>(aux)[j0]; t.a1 = reinterpret_cast<const f4*>(aux)[j1]; }
        }
        return t;
    };
    Trip cur = fetch((int64_t)blockIdx.x * 256 + threadIdx.x);
    hist_zero(h);
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
    // the error bound of the half-bytes search for this query (the statements of approx_threshold_kernel: same bits), for the filter and
    // the final kernel of the selection that follows
    if (hb.m_out && blockIdx.x == 0 && threadIdx.x == 0)
        hb.m_out[b] = mode == SCAN_COSINE ? hb.m_rel : mode == SCAN_L2 ? l2_delta(hb.e_norm_bound, hb.m_rel, hb.e_max, qn) : hb.m_rel * hb.e_norm_bound * qn + 0x1p-22f;
    auto one = [&](int64_t i) {  // scalar tail / unaligned layout
        const float o = transform_score(sb[i] * pre_scale, mode, mode == SCAN_COSINE ? row_norm[i] : 1.f, mode == SCAN_L2 ? row_sumsq[i] : 0.f, qn, qss);
        hist_add(h, o);
        sb[i] = o;
    };
    auto finish = [&](int64_t i, const f4 d, const f4 a) {
        f4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u] = transform_score(d[u] * pre_scale, mode, a[u], a[u], qn, qss);  // a = the row norms (cosine) or squared norms (l2)
            hist_add(h, o[u]);
        }
        reinterpret_cast<f4*>(sb)[i] = o;
    };
    if (vec) {
        while (cur.has0) {
            const Trip nxt = fetch(cur.i0 + 2 * stride);
            finish(cur.i0, cur.d0, cur.a0);
            if (cur.has1) finish(cur.i1, cur.d1, cur.a1);
            cur = nxt;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) one((n4 << 2) + threadIdx.x);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) one(i);
    }
    __syncthreads();
    hist_flush(h, ws_hist + (int64_t)b * HIST_STRIDE);
}

// ---- 2. filter ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_filter_kernel(const float* __restrict__ scores, int64_t n, int64_t ld,
                                                           int32_t k, uint32_t* __restrict__ ws_hist,
                                                           uint64_t* __restrict__ ws_sel,
                                                           uint64_t* __restrict__ ws_cand, const uint32_t* __restrict__ run_if,
                                                           const float* __restrict__ below_m, int below_l2) {
    __shared__ uint32_t h[HIST_BINS];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t thr[2];
    if (run_if && *run_if == 0u) return;
    const int q = blockIdx.y;
    uint32_t* g = ws_hist + (int64_t)q * HIST_STRIDE;
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) h[i] = g[i];
    __syncthreads();
    const uint32_t need = (uint32_t)std::min<int64_t>(k, n);
    find_threshold_bin(h, HIST_BINS, need, scratch, thr);
    const uint32_t bstar = thr[0];
    const float* s = scores + (int64_t)q * ld;
    uint64_t* sel = ws_sel + (int64_t)q * K_MAX;
    uint64_t* cand = ws_cand + (int64_t)q * CAND_CAP;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // below_m != nullptr (the half-bytes row search, api.hip: search_rows_hi): scores less than 2 m below the threshold bin's lower edge
    // are candidates too -- the final kernel then finds every score within 2 m of the k-th best among sel + cand (the k-th best lies IN the
    // bin), and no separate pass over the scores has to collect them.  They rank below the whole bin: the selection itself is unchanged.
    const float low = below_m ? lower_threshold(key_score(bstar << 21), 1.0001f * below_m[q], below_l2 != 0) : INFINITY;  // (the bin's edge bounds the k-th best from below)
    auto visit = [&](float v, int64_t i) {
        const uint32_t bin = score_key(v) >> 21;
        if (bin > bstar) {
            const uint32_t p = atomicAdd(&g[CNT_SEL], 1u);
            if (p < (uint32_t)K_MAX) sel[p] = make_key64(v, (uint32_t)i);
        } else if (bin == bstar || v >= low) {
            const uint32_t p = atomicAdd(&g[CNT_CAND], 1u);
            if (p < (uint32_t)CAND_CAP) cand[p] = make_key64(v, (uint32_t)i);
        }
    };
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0) {  // 16-B loads, streamed once more: nt
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4* s4 = reinterpret_cast<const f4*>(s);
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const f4 v = __builtin_nontemporal_load(s4 + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) visit(v[u], (i << 2) + u);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) visit(s[(n4 << 2) + threadIdx.x], (n4 << 2) + threadIdx.x);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) visit(s[i], i);
    }
}

// ---- 3. final ----------------------------------------------------------------------------------------
__device__ __forceinline__ void write_results(const uint64_t* sorted, int n_valid, int32_t k,
                                              float* __restrict__ out_scores, int32_t* __restrict__ out_ids) {
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint64_t key = (i < n_valid) ? sorted[i] : 0ull;
        if (key == 0ull) {
            out_scores[i] = -INFINITY;
            out_ids[i] = -1;
        } else {
            const uint32_t k32 = (uint32_t)(key >> 32);
            out_scores[i] = k32 ? key_score(k32) : __uint_as_float(0x7fc00000u);
            out_ids[i] = (int32_t)(0xffffffffu - (uint32_t)key);
        }
    }
}

// The `need` best keys among the scores whose key falls into bin `bstar` (any number of them), by one block: two more radix passes over the
// scores fix the exact 32-bit threshold, a third takes what lies above it and the ties on it in index order -- slow but exact.  Results
// go to fin[n_sel .. n_sel + need).  h: HIST_BINS words of LDS, scratch >= 20, thr 2, sh_cnt 3.  Every thread of the block must call.
__device__ __forceinline__ void refine_in_bin(const float* __restrict__ s, int64_t n, uint32_t bstar, uint32_t need, uint32_t n_sel,
                                              uint64_t* fin, uint32_t* h, uint32_t* scratch, uint32_t* thr, uint32_t* sh_cnt) {
    __syncthreads();
    // pass A: bits [20:10] among keys in bin b*
    for (int i = threadIdx.x; i < HIST_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t key = score_key(s[i]);
        if ((key >> 21) == bstar) atomicAdd(&h[(key >> 10) & 2047u], 1u);
    }
    __syncthreads();
    find_threshold_bin(h, HIST_BINS, need, scratch, thr);
    const uint32_t b1 = thr[0];
    const uint32_t need1 = need - thr[1];
    __syncthreads();
    // pass B: bits [9:0] among keys matching (b*, b1)
    for (int i = threadIdx.x; i < HIST_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t prefix22 = (bstar << 11) | b1;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t key = score_key(s[i]);
        if ((key >> 10) == prefix22) atomicAdd(&h[key & 1023u], 1u);
    }
    __syncthreads();
    find_threshold_bin(h, 1024, need1, scratch, thr);
    const uint32_t t32 = (prefix22 << 10) | thr[0];
    const uint32_t need_eq = need1 - thr[1];  // ties on the exact threshold value still to take
    __syncthreads();
    // pass C: keys in bin b* above t32 (any order) + the first `need_eq` keys == t32 by index.
    if (threadIdx.x == 0) { sh_cnt[0] = 0; sh_cnt[1] = 0; }
    __syncthreads();
    for (int64_t base = 0; base < n; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        uint32_t key = 0;
        float v = 0.f;
        if (i < n) { v = s[i]; key = score_key(v); }
        const bool gt = (i < n) && ((key >> 21) == bstar) && (key > t32);
        const bool eq = (i < n) && (key == t32);
        if (gt) {
            const uint32_t p = atomicAdd(&sh_cnt[0], 1u);
            fin[n_sel + p] = make_key64(v, (uint32_t)i);
        }
        uint32_t tot;
        const uint32_t incl = block_inclusive_scan(eq ? 1u : 0u, scratch, tot);
        const uint32_t rank = sh_cnt[1] + incl - 1;  // 0-based rank of this tie in index order
        __syncthreads();
        if (eq && rank < need_eq) fin[n_sel + (need - need_eq) + rank] = make_key64(v, (uint32_t)i);
        // The exit decision is taken by ONE thread and published in LDS: every thread reading the counters itself would
        // race with a faster wave already counting the next iteration's `gt` hits (a non-uniform break leaves waves
        // behind at the scan's barriers).
        if (threadIdx.x == 0) {
            sh_cnt[1] += tot;
            sh_cnt[2] = (sh_cnt[1] >= need_eq && sh_cnt[0] >= need - need_eq) ? 1u : 0u;
        }
        __syncthreads();
        const bool done = sh_cnt[2] != 0;
        __syncthreads();  // nobody writes sh_cnt again before everybody has read the flag
        if (done) break;
    }
    __syncthreads();
}

// LDS of the final step (56 KiB).
struct FinalLds {
    uint64_t buf[CAND_CAP];   // 32 KiB: candidate sort, then reused as the result sort buffer
    uint64_t fin[K_MAX];      // 16 KiB
    uint32_t h[HIST_BINS];    // 8 KiB
    uint32_t scratch[20];
    uint32_t thr[2];
    uint32_t sh_cnt[3];
};

// The final step for query q, by one block of 1024 threads (every thread of the block must call).
__device__ __forceinline__ void topk_final_body(const float* __restrict__ scores, int64_t n, int64_t ld, int32_t k,
                                                uint32_t* __restrict__ ws_hist, const uint64_t* __restrict__ ws_sel,
                                                const uint64_t* __restrict__ ws_cand, float* __restrict__ out_scores,
                                                int32_t* __restrict__ out_ids, int q, FinalLds& L, const HiEmit& em = HiEmit{}) {
    uint64_t* const buf = L.buf;
    uint64_t* const fin = L.fin;
    uint32_t* const h = L.h;
    uint32_t* const scratch = L.scratch;
    uint32_t* const thr = L.thr;
    uint32_t* const sh_cnt = L.sh_cnt;
    uint32_t* g = ws_hist + (int64_t)q * HIST_STRIDE;
    const uint64_t* sel = ws_sel + (int64_t)q * K_MAX;
    const uint64_t* cand = ws_cand + (int64_t)q * CAND_CAP;
    float* os = out_scores + (int64_t)q * k;
    int32_t* oi = out_ids + (int64_t)q * k;
    const uint32_t kk = (uint32_t)std::min<int64_t>(k, n);
    const uint32_t n_sel = g[CNT_SEL];
    const uint32_t n_cand = g[CNT_CAND];
    const uint32_t need = kk - n_sel;  // >= 1 whenever kk >= 1 (the threshold bin is never empty)
    // em.ids != nullptr (the half-bytes row search): every score within 2 m of the k-th best is a candidate for exact re-scoring -- all of
    // sel (above the k-th best) and the part of cand (threshold bin + the filter's margin below it) that reaches thr = (k-th best) - 2 m.
    // Their rows go to em.ids[q * cap ..] (any order; with em.row_norm their norms next to them), em.cnt[q] = how many; more than cap,
    // fewer than k rows, a candidate list the filter could not hold, or an unusable threshold: *em.flag (the guarded full pass answers).
    const bool emit = em.ids != nullptr;
    auto emit_key = [&](uint64_t key, float t) {
        const uint32_t k32 = (uint32_t)(key >> 32);
        if (key == 0ull || k32 == 0u || !(key_score(k32) >= t)) return;
        const int32_t row = (int32_t)(0xffffffffu - (uint32_t)key);
        const uint32_t p = atomicAdd(&sh_cnt[0], 1u);  // (this block is the only writer of the query's list: the counter lives in LDS)
        if (p < (uint32_t)em.cap) {
            em.ids[(int64_t)q * em.cap + p] = row;
            if (em.row_norm) em.norms[(int64_t)q * em.cap + p] = em.row_norm[row];
        }
    };
    auto emit_done = [&]() {  // (after a barrier behind the last emit_key)
        if (threadIdx.x == 0) {
            em.cnt[q] = sh_cnt[0];
            if (sh_cnt[0] > (uint32_t)em.cap) atomicOr(em.flag, 1u);
        }
    };
    if (emit && threadIdx.x == 0) sh_cnt[0] = 0u;  // (ordered before the first emit_key by the barriers below)
    auto emit_threshold = [&](uint64_t kth_key) -> float {  // (every thread computes the same value)
        const float t = lower_threshold(key_score((uint32_t)(kth_key >> 32)), em.m[q], em.l2 != 0);
        if (threadIdx.x == 0) {
            if (em.thr) em.thr[q] = t;
            if (!(t > -INFINITY) || kk < (uint32_t)k) atomicOr(em.flag, 1u);  // NaN / -inf, or fewer than k rows
        }
        return t;
    };
    // This block is the last reader of the query's histogram row: keep the bins (the slow path wants them) and hand the row
    // back all zero, which is what the next selection's histogram pass expects (no memset launch per top-k).
    for (int i = threadIdx.x; i < HIST_BINS; i += blockDim.x) h[i] = g[i];
    for (int i = threadIdx.x; i < K_MAX; i += blockDim.x) fin[i] = (i < (int)n_sel) ? sel[i] : 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_STRIDE; i += blockDim.x) g[i] = 0u;
    if (kk == 0) { write_results(fin, 0, k, os, oi); return; }

    if (n_cand <= (uint32_t)RANK_MAX) {
        // Few candidates (the common case: a few hundred): every thread ranks one key by counting the larger ones
        // (all keys are distinct; the LDS reads are wave-wide broadcasts) -- two barriers instead of a sorting network.
        for (int i = threadIdx.x; i < (int)n_cand; i += blockDim.x) buf[i] = cand[i];
        __syncthreads();
        if (threadIdx.x < n_cand) {
            const uint64_t mine = buf[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n_cand; ++j) rank += buf[j] > mine;
            if (rank < need) fin[n_sel + rank] = mine;
        }
        __syncthreads();
        if (emit) {  // fin[kk - 1] is the k-th best (the need-th of cand); buf still holds cand
            const float t = emit_threshold(fin[kk - 1]);
            if (threadIdx.x < n_cand) emit_key(buf[threadIdx.x], t);
            for (int i = threadIdx.x; i < (int)n_sel; i += blockDim.x) emit_key(fin[i], t);
            __syncthreads();
            emit_done();
        }
    } else if (n_cand <= (uint32_t)CAND_CAP) {
        int p2 = 64;
        while (p2 < (int)n_cand) p2 <<= 1;
        for (int i = threadIdx.x; i < p2; i += blockDim.x) buf[i] = (i < (int)n_cand) ? cand[i] : 0ull;
        __syncthreads();
        bitonic_sort_desc(buf, p2);
        for (int i = threadIdx.x; i < (int)need; i += blockDim.x) fin[n_sel + i] = buf[i];
        __syncthreads();
        if (emit) {
            const float t = emit_threshold(buf[need - 1]);
            for (int i = threadIdx.x; i < (int)n_cand; i += blockDim.x) emit_key(buf[i], t);
            for (int i = threadIdx.x; i < (int)n_sel; i += blockDim.x) emit_key(fin[i], t);
            __syncthreads();
            emit_done();
        }
    } else {
        if (emit && threadIdx.x == 0) atomicOr(em.flag, 1u);  // (more candidates than the filter's list holds: no emission from it)
        // Slow exact path: refine the 32-bit threshold inside bin b*, then take ties in index order.
        // (Recompute b* from the histogram -- copied to LDS above -- exactly as the filter kernel did.)
        find_threshold_bin(h, HIST_BINS, kk, scratch, thr);
        refine_in_bin(scores + (int64_t)q * ld, n, thr[0], need, n_sel, fin, h, scratch, thr, sh_cnt);
    }
    // Order the kk survivors (all distinct keys) and write them out.
    if (kk <= (uint32_t)RANK_MAX) {
        if (threadIdx.x < kk) {
            const uint64_t mine = fin[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < kk; ++j) rank += fin[j] > mine;
            buf[rank] = mine;
        }
        __syncthreads();
        write_results(buf, (int)kk, k, os, oi);
        return;
    }
    int p2 = 64;
    while (p2 < (int)kk) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += blockDim.x) buf[i] = (i < (int)kk) ? fin[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(buf, p2);
    write_results(buf, (int)kk, k, os, oi);
}

__global__ __launch_bounds__(1024) void topk_final_kernel(const float* __restrict__ scores, int64_t n, int64_t ld,
                                                           int32_t k, uint32_t* __restrict__ ws_hist,
                                                           const uint64_t* __restrict__ ws_sel,
                                                           const uint64_t* __restrict__ ws_cand,
                                                           float* __restrict__ out_scores,
                                                           int32_t* __restrict__ out_ids, const uint32_t* __restrict__ run_if, HiEmit em) {
    __shared__ FinalLds L;
    if (run_if && *run_if == 0u) return;
    topk_final_body(scores, n, ld, k, ws_hist, ws_sel, ws_cand, out_scores, out_ids, (int)blockIdx.x, L, em);
}

// ---- round 6: the whole selection of a query in ONE launch, one block per query ("block route") ------------------------------------------
// hist -> filter -> final is three launches, a global histogram, two global candidate lists and a sort of the threshold bin (a MaxSim score of
// ~700 falls into the bin [704, 768) with a few thousand others: 45 us of bitonic network per step of the headline pipeline, 113 us for the
// three kernels; the sample of the fused row top-k pays 100 us per search for the same).  When a query's scores are few enough to be read
// twice by ONE block out of L2 / the Infinity Cache -- n <= 256 k: MaxSim chunk scores, the fused top-k's sample, rerank lists -- the block does
// it all by itself: (1) histogram of the key's top 11 bits in LDS (four lane-interleaved copies), threshold bin b*; (2) second read: keys
// above b* go to the result list, the keys IN b* to an LDS buffer; (3) inside the buffer a second radix level (key bits 20..10) leaves a
// handful of keys on the threshold value, which are ranked by counting; (4) the k survivors are ordered and written.  No workspace, no
// global atomics, the same (score desc, id asc) order on the same unique 64-bit keys: bit-identical results.  More keys in b* than the buffer
// holds, or more than 1024 on one 22-bit prefix (massive ties): refine_in_bin over the global scores, the exact slow path of the final kernel.
//
// PREFILTER (round 6, option topk_block = 2, the default; k <= 512): most of that work is spent on scores that cannot matter -- a MaxSim query's
// 125 k chunk scores crowd into two or three bins, whose LDS atomics serialise.  Every thread first takes the MAXIMUM key of the elements it
// reads (thread t: 16-byte groups t, t + 1024, ...; no atomics); the k-th largest of the 1024 thread maxima -- 1024 DISTINCT elements, ordered by
// one bitonic sort in LDS -- is a lower bound P of the k-th best key overall, and the keys >= P are the k thread maxima above it plus what else
// their threads hold above P: about -1024 ln(1 - k / 1024) keys (105 at k = 100).  A second read collects them (one ballot per 16-byte group,
// an append only where a lane has one), they are ranked, the best k written.  The keys are unique (score bits, ~index), so ties cost nothing.
// Only data whose large keys all sit in few threads' groups can overflow the buffer (more than 8192 keys >= P): the histogram path
// below then answers, as it does for k > 512.  Same keys, same order: bit-identical results (tests/test_gpu_topk_block.py).
constexpr int BLOCK_BUF = 8192;          // keys of the threshold bin kept in LDS
constexpr int64_t BLOCK_ROUTE_MAX_N = 262144;
constexpr int PREFILTER_MAX_K = 512;
struct BlockLds {
    uint64_t buf[BLOCK_BUF];                      // 64 KiB: the threshold bin; later the sort buffer of the results
    uint64_t fin[K_MAX];                          // 16 KiB
    uint64_t tie[RANK_MAX];                       //  8 KiB: keys on the threshold sub-bin (prefilter: the thread maxima)
    uint32_t h[HIST_COPIES * HIST_COPY_STRIDE];   // 33 KiB
    uint32_t scratch[20];
    uint32_t thr[2];
    uint32_t sh_cnt[4];
    uint64_t pivot;
};
// position of this lane's element in a list that `pred` lanes of the wave append to (one LDS atomic per wave)
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
    const uint64_t mask = __builtin_amdgcn_ballot_w64(pred);
    if (mask == 0ull) return 0u;
    uint32_t base = 0u;
    if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(counter, (uint32_t)__builtin_popcountll(mask));
    base = __builtin_amdgcn_readlane(base, __builtin_ctzll(mask));
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__global__ __launch_bounds__(1024) void topk_block_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, int32_t k,
                                                           float* __restrict__ out_scores, int32_t* __restrict__ out_ids, int prefilter) {
    __shared__ BlockLds L;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int q = blockIdx.x;
    const float* const s = scores + (int64_t)q * ld;
    float* const os = out_scores + (int64_t)q * k;
    int32_t* const oi = out_ids + (int64_t)q * k;
    const uint32_t kk = (uint32_t)std::min<int64_t>(k, n);
    if (kk == 0) { write_results(L.fin, 0, k, os, oi); return; }
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(scores) & 15) == 0;
    const int64_t n4 = vec ? n >> 2 : 0;
    if (prefilter && kk <= (uint32_t)PREFILTER_MAX_K) {
        const f4* const s4 = reinterpret_cast<const f4*>(s);
        const f4 none = (f4){NAN, NAN, NAN, NAN};  // a group past the end: key bits 0, never a maximum
        constexpr int REG_GROUPS = 12;             // up to 12 x 1024 16-byte groups (49 152 scores: the fused top-k's sample) stay in registers: ONE read
        const bool resident = n4 > 0 && n4 <= (int64_t)REG_GROUPS * 1024 && (n & 3) == 0;  // (block-uniform)
        f4 r[REG_GROUPS];
        // (a) ONE of this thread's best elements (a float compare: the first of equal scores stays -- on massive ties the representatives are
        // then the lowest indices, which is the selection's own order; -0 / +0 count as equal, NaN and -inf never win: ANY element per thread
        // keeps the pivot a valid lower bound, a better one only makes it tighter)
        float bv = -INFINITY;
        uint32_t bi = 0u;
        auto take = [&](float v, int64_t i) {
            if (v > bv) { bv = v; bi = (uint32_t)i; }
        };
        if (resident) {
#pragma unroll
            for (int j = 0; j < REG_GROUPS; ++j) {
                const int64_t i = (int64_t)threadIdx.x + 1024 * j;
                r[j] = i < n4 ? s4[i] : none;
            }
#pragma unroll
            for (int j = 0; j < REG_GROUPS; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) take(r[j][u], (((int64_t)threadIdx.x + 1024 * j) << 2) + u);
        } else {
            for (int64_t i = threadIdx.x; i < n4; i += 8192) {  // eight 16-byte loads in flight per lane: the block is alone on its CU, and a
                f4 v[8];                                        // round trip to the cache the pass kernel left the scores in is ~1 us
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = i + 1024 * j < n4 ? s4[i + 1024 * j] : none;
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int u = 0; u < 4; ++u) take(v[j][u], ((i + 1024 * j) << 2) + u);
            }
            for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) take(s[i], i);
        }
        // (a thread without elements, or with NaN / -inf only, has no representative: key 0 sorts last)
        uint64_t mine = bv > -INFINITY ? make_key64(bv, bi) : 0ull;
        // k <= 128: the maxima of the 256 lane QUADS (~ -256 ln(1 - k / 256) keys reach their k-th largest: 127 at k = 100), ranked by counting by
        // four waves; beyond: the 1024 thread maxima through a bitonic sort (55 barriers)
        const bool quads = kk <= 128u;
        if (quads) {
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)mine, o, 64), hi = __shfl_xor((uint32_t)(mine >> 32), o, 64);
                const uint64_t other = ((uint64_t)hi << 32) | lo;
                mine = other > mine ? other : mine;
            }
            if ((threadIdx.x & 3) == 0) L.tie[threadIdx.x >> 2] = mine;
        } else {
            L.tie[threadIdx.x] = mine;
        }
        if (threadIdx.x == 0) { L.sh_cnt[0] = 0u; L.pivot = 0ull; }
        __syncthreads();
        if (quads) {
            {   // all sixteen waves: lane quad t >> 2 ranks key t >> 2, each of its lanes against a quarter of the keys
                const uint64_t m = L.tie[threadIdx.x >> 2];
                uint32_t rank = 0;
                for (int j = (threadIdx.x & 3); j < 256; j += 4) rank += L.tie[j] > m;
                rank += __shfl_xor(rank, 1, 64);
                rank += __shfl_xor(rank, 2, 64);
                if ((threadIdx.x & 3) == 0 && rank == kk - 1u) L.pivot = m;  // (distinct keys; quads without one share key 0 and rank behind all others)
            }
            __syncthreads();
        } else {
            bitonic_sort_desc(L.tie, RANK_MAX);
            if (threadIdx.x == 0) L.pivot = L.tie[kk - 1];
            __syncthreads();
        }
        const uint64_t pivot = L.pivot;  // 0: fewer than kk maxima -- every key passes
        const bool all = pivot == 0ull;
        const float pf = key_score((uint32_t)(pivot >> 32));  // coarse test: a float compare (never misses a key >= pivot; the key decides)
        // (b) the keys >= pivot
        auto collect = [&](float v, int64_t i, bool coarse) {
            const uint64_t key = make_key64(v, (uint32_t)i);
            const bool hit = coarse && key >= pivot;
            const uint32_t pos = wave_append(&L.sh_cnt[0], hit);
            if (hit && pos < (uint32_t)BLOCK_BUF) L.buf[pos] = key;
        };
        auto group = [&](const f4 v, int64_t g, bool in) {  // (called by whole waves: the ballots are wave-wide)
            bool c[4];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = in && (all || v[u] >= pf); any |= c[u]; }
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;  // (wave-uniform)
#pragma unroll
            for (int u = 0; u < 4; ++u) collect(v[u], (g << 2) + u, c[u]);
        };
        if (resident) {
#pragma unroll
            for (int j = 0; j < REG_GROUPS; ++j) group(r[j], (int64_t)threadIdx.x + 1024 * j, (int64_t)threadIdx.x + 1024 * j < n4);
        } else {
            for (int64_t i0 = 0; i0 < n4; i0 += 4096) {
                const int64_t i = i0 + threadIdx.x;
                f4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = i + 1024 * j < n4 ? s4[i + 1024 * j] : none;
#pragma unroll
                for (int j = 0; j < 4; ++j) group(v[j], i + 1024 * j, i + 1024 * j < n4);
            }
            for (int64_t i0 = n4 << 2; i0 < n; i0 += 1024) {
                const int64_t i = i0 + threadIdx.x;
                const float v = i < n ? s[i] : 0.f;
                collect(v, i, i < n && (all || v >= pf));
            }
        }
        __syncthreads();
        const uint32_t c_n = L.sh_cnt[0];  // >= kk (kk maxima >= pivot are among them; pivot 0: all n keys)
        if (c_n <= (uint32_t)BLOCK_BUF) {
            if (c_n <= 256u) {  // ranked by counting (distinct keys), a lane quad per key
                const uint32_t a = threadIdx.x >> 2;
                const uint64_t m = a < c_n ? L.buf[a] : 0ull;
                uint32_t rank = 0;
                for (uint32_t j = (threadIdx.x & 3); j < c_n; j += 4) rank += L.buf[j] > m;
                rank += __shfl_xor(rank, 1, 64);
                rank += __shfl_xor(rank, 2, 64);
                if ((threadIdx.x & 3) == 0 && a < c_n && rank < kk) L.fin[rank] = m;
                __syncthreads();
                write_results(L.fin, (int)kk, k, os, oi);
            } else {
                int p2 = 512;
                while (p2 < (int)c_n) p2 <<= 1;
                for (int i = c_n + threadIdx.x; i < p2; i += 1024) L.buf[i] = 0ull;
                __syncthreads();
                bitonic_sort_desc(L.buf, p2);
                write_results(L.buf, (int)kk, k, os, oi);
            }
            return;
        }
        __syncthreads();  // (overflow: the exact histogram path below, from scratch)
    }
    // (1) histogram of the top 11 key bits
    hist_zero(L.h);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
        const f4 v = reinterpret_cast<const f4*>(s)[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) hist_add(L.h, v[u]);
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) hist_add(L.h, s[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 1024) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < HIST_COPIES; ++j) c += L.h[j * HIST_COPY_STRIDE + i];
        L.h[i] = c;  // (copy 0 now holds the sums: bin i of copy 0 is read and written by this thread only)
    }
    if (threadIdx.x < 4) L.sh_cnt[threadIdx.x] = 0u;
    __syncthreads();
    find_threshold_bin(L.h, HIST_BINS, kk, L.scratch, L.thr);
    const uint32_t bstar = L.thr[0];
    // (2) keys above b* -> fin, keys in b* -> buf
    auto visit = [&](float v, int64_t i, bool live) {
        const uint32_t bin = score_key(v) >> 21;
        const bool above = live && bin > bstar, in_bin = live && bin == bstar;
        const uint32_t pa = wave_append(&L.sh_cnt[0], above);
        if (above) L.fin[pa] = make_key64(v, (uint32_t)i);  // (fewer than kk <= K_MAX keys lie above b*)
        const uint32_t pb = wave_append(&L.sh_cnt[1], in_bin);
        if (in_bin && pb < (uint32_t)BLOCK_BUF) L.buf[pb] = make_key64(v, (uint32_t)i);
    };
    for (int64_t i0 = 0; i0 < n4; i0 += 1024) {  // (whole waves stay in the loop: wave_append is wave-wide)
        const int64_t i = i0 + threadIdx.x;
        const bool live = i < n4;
        f4 v = (f4){0.f, 0.f, 0.f, 0.f};
        if (live) v = reinterpret_cast<const f4*>(s)[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) visit(v[u], (i << 2) + u, live);
    }
    for (int64_t i0 = n4 << 2; i0 < n; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        visit(i < n ? s[i] : 0.f, i, i < n);
    }
    __syncthreads();
    const uint32_t n_sel = L.sh_cnt[0], n_bin = L.sh_cnt[1];
    const uint32_t need = kk - n_sel;  // >= 1: the threshold bin is never empty
    __syncthreads();
    if (n_bin <= (uint32_t)RANK_MAX) {
        // a small bin: every thread ranks one key by counting the larger ones (all keys are distinct)
        if (threadIdx.x < n_bin) {
            const uint64_t mine = L.buf[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n_bin; ++j) rank += L.buf[j] > mine;
            if (rank < need) L.fin[n_sel + rank] = mine;
        }
        __syncthreads();
    } else if (n_bin <= (uint32_t)BLOCK_BUF) {
        // (3) second radix level inside the buffer: key bits 20..10
        for (int i = threadIdx.x; i < HIST_BINS; i += 1024) L.h[i] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_bin; i += 1024) atomicAdd(&L.h[(uint32_t)(L.buf[i] >> 42) & 2047u], 1u);
        __syncthreads();
        find_threshold_bin(L.h, HIST_BINS, need, L.scratch, L.thr);
        const uint32_t b1 = L.thr[0], above1 = L.thr[1];
        const uint32_t n_tie = L.h[b1];
        __syncthreads();
        if (n_tie <= (uint32_t)RANK_MAX) {
            for (uint32_t i0 = 0; i0 < n_bin; i0 += 1024) {
                const uint32_t i = i0 + threadIdx.x;
                const bool live = i < n_bin;
                const uint64_t key = live ? L.buf[i] : 0ull;
                const uint32_t sub = (uint32_t)(key >> 42) & 2047u;
                const uint32_t pa = wave_append(&L.sh_cnt[2], live && sub > b1);
                if (live && sub > b1) L.fin[n_sel + pa] = key;
                const uint32_t pt = wave_append(&L.sh_cnt[3], live && sub == b1);
                if (live && sub == b1) L.tie[pt] = key;
            }
            __syncthreads();
            const uint32_t need1 = need - above1;  // of the n_tie keys on the threshold sub-bin
            if (threadIdx.x < n_tie) {
                const uint64_t mine = L.tie[threadIdx.x];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n_tie; ++j) rank += L.tie[j] > mine;
                if (rank < need1) L.fin[n_sel + above1 + rank] = mine;
            }
            __syncthreads();
        } else {
            refine_in_bin(s, n, bstar, need, n_sel, L.fin, L.h, L.scratch, L.thr, L.sh_cnt);
        }
    } else {
        refine_in_bin(s, n, bstar, need, n_sel, L.fin, L.h, L.scratch, L.thr, L.sh_cnt);
    }
    // (4) order the kk survivors (all distinct keys) and write them out
    if (kk <= (uint32_t)RANK_MAX) {
        if (threadIdx.x < kk) {
            const uint64_t mine = L.fin[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < kk; ++j) rank += L.fin[j] > mine;
            L.buf[rank] = mine;
        }
        __syncthreads();
        write_results(L.buf, (int)kk, k, os, oi);
        return;
    }
    int p2 = 64;
    while (p2 < (int)kk) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += 1024) L.buf[i] = (i < (int)kk) ? L.fin[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(L.buf, p2);
    write_results(L.buf, (int)kk, k, os, oi);
}

// ---- the guarded fallback of the half-bytes row search in ONE launch -------------------------------------------------------------------
// Raw dots -> similarities (in place; transform_kernel's statements) and their exact top-k, by ONE block per query: histogram of the key's
// top 11 bits in LDS, the keys above the threshold bin, then refine_in_bin for the bin itself.  ~1 ms per query over 1 M scores -- it runs
// only when *run_if != 0 (a candidate list overflowed, an unusable bound); the point is that the launch that usually returns at once is
// ONE launch instead of three (transform + histogram, filter, final: 14 us of a 0.38 ms single-query search).
__global__ __launch_bounds__(1024) void guarded_select_kernel(float* __restrict__ scores, int64_t n, int64_t ld, int32_t k,
                                                               const float* __restrict__ row_norm, const float* __restrict__ row_sumsq,
                                                               const float* __restrict__ queries, int dim, int mode, float pre_scale,
                                                               float* __restrict__ out_scores, int32_t* __restrict__ out_ids,
                                                               const uint32_t* __restrict__ run_if, uint32_t* __restrict__ host_flag) {
    __shared__ FinalLds L;
    __shared__ float part[4];
    // host_flag: a word of pinned host memory that learns whether the guarded path ran (lazy images: the next batch then asks for the pre-split
    // image) -- written by the kernel that reads the flag anyway instead of a 4-byte device-to-host copy behind it (5 us per batch)
    if (host_flag && run_if && blockIdx.x == 0 && threadIdx.x == 0) *host_flag = *run_if;
    if (run_if && *run_if == 0u) return;
    const int q = blockIdx.x;
    float* const sb = scores + (int64_t)q * ld;
    if (threadIdx.x < 256) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < dim; c += 256) {
            const float v = queries[(int64_t)q * dim + c];
            ss = fmaf(v, v, ss);
        }
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    }
    for (int i = threadIdx.x; i < HIST_BINS; i += blockDim.x) L.h[i] = 0u;
    for (int i = threadIdx.x; i < K_MAX; i += blockDim.x) L.fin[i] = 0ull;
    if (threadIdx.x == 0) L.sh_cnt[0] = 0u;
    __syncthreads();
    const float qss = (part[0] + part[1]) + (part[2] + part[3]);
    const float qn = sqrtf(qss);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float o = transform_score(sb[i] * pre_scale, mode, mode == SCAN_COSINE ? row_norm[i] : 1.f, mode == SCAN_L2 ? row_sumsq[i] : 0.f, qn, qss);
        sb[i] = o;
        atomicAdd(&L.h[score_key(o) >> 21], 1u);
    }
    __syncthreads();
    const uint32_t kk = (uint32_t)std::min<int64_t>(k, n);
    float* os = out_scores + (int64_t)q * k;
    int32_t* oi = out_ids + (int64_t)q * k;
    if (kk == 0) { write_results(L.fin, 0, k, os, oi); return; }
    find_threshold_bin(L.h, HIST_BINS, kk, L.scratch, L.thr);
    const uint32_t bstar = L.thr[0], n_sel = L.thr[1];
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {  // the keys above the threshold bin (fewer than kk), any order
        const float v = sb[i];
        if ((score_key(v) >> 21) > bstar) L.fin[atomicAdd(&L.sh_cnt[0], 1u)] = make_key64(v, (uint32_t)i);
    }
    __syncthreads();
    refine_in_bin(sb, n, bstar, kk - n_sel, n_sel, L.fin, L.h, L.scratch, L.thr, L.sh_cnt);
    int p2 = 64;
    while (p2 < (int)kk) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += blockDim.x) L.buf[i] = (i < (int)kk) ? L.fin[i] : 0ull;
    __syncthreads();
    bitonic_sort_desc(L.buf, p2);
    write_results(L.buf, (int)kk, k, os, oi);
}

int launch_guarded_select(float* scores, int32_t nb, int64_t n, int64_t ld, int32_t k, const float* row_norm, const float* row_sumsq,
                          const float* queries, int32_t dim, int mode, float pre_scale, float* out_scores, int32_t* out_ids,
                          const uint32_t* run_if, hipStream_t s, uint32_t* host_flag) {
    if (nb <= 0 || k <= 0) return RL_OK;
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "top-k: k must be <= 2048");
    if (n >= (int64_t)0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "top-k: more than 2^31-2 elements per query");
    hipLaunchKernelGGL(guarded_select_kernel, dim3(nb), dim3(1024), 0, s, scores, n, ld, k, row_norm, row_sumsq, queries, (int)dim, mode, pre_scale,
                       out_scores, out_ids, run_if, host_flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int select_workspace_reserve(SelectWorkspace& ws, int32_t nq, hipStream_t s) {
    if (nq > ws.capacity_queries) {
        select_workspace_free(ws);
        RL_HIP(hipMalloc(&ws.hist, (size_t)nq * HIST_STRIDE * sizeof(uint32_t)));
        RL_HIP(hipMalloc(&ws.sel, (size_t)nq * K_MAX * sizeof(uint64_t)));
        RL_HIP(hipMalloc(&ws.cand, (size_t)nq * CAND_CAP * sizeof(uint64_t)));
        RL_HIP(hipMalloc(&ws.pv, ((size_t)nq * 3 + 16) * sizeof(uint32_t)));
        ws.capacity_queries = nq;
        ws.dirty = true;
    }
    if (ws.dirty) {  // fresh allocation, or an earlier launch sequence did not get to its final kernel
        RL_HIP(hipMemsetAsync(ws.hist, 0, (size_t)ws.capacity_queries * HIST_STRIDE * sizeof(uint32_t), s));
        ws.dirty = false;
    }
    return RL_OK;
}

void select_workspace_free(SelectWorkspace& ws) {
    if (ws.hist) (void)hipFree(ws.hist);
    if (ws.sel) (void)hipFree(ws.sel);
    if (ws.cand) (void)hipFree(ws.cand);
    if (ws.pv) (void)hipFree(ws.pv);
    ws = SelectWorkspace{};
}

int launch_topk_pivot(const float* scores, int32_t nq, int64_t n, int64_t ld, int32_t k, SelectWorkspace& ws, float* out_scores, int32_t* out_ids,
                      hipStream_t s) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (k > 512 || nq > 240 || n >= (int64_t)0x7fffffff || !pivot_route_takes(n, k)) return RL_ERR_UNSUPPORTED;  // (k <= 512 since the pivot takes G <= 2048 groups)
    const int block_route = ws.block_route;
    RL_TRY(select_workspace_reserve(ws, nq, s));
    ws.block_route = block_route;
    // the selection's own buffers, idle on this route: group maxima in `sel` (up to all 2048 of its words per query), the collected (id, score) lists in
    // `cand` (4096 + 4096 four-byte words per query = its 4096 eight-byte ones)
    constexpr int32_t cap = CAND_CAP;
    uint64_t* bmax = ws.sel;
    int32_t* ids = reinterpret_cast<int32_t*>(ws.cand);
    float* vals = reinterpret_cast<float*>(ids + (size_t)nq * cap);
    uint32_t* cnt = ws.pv;
    uint32_t* flag = cnt + nq;
    float* m = reinterpret_cast<float*>(flag + 16);
    float* thr = m + nq;
    HiBound bound;
    bound.m_out = m;  // (no error band here: the kernel leaves 2^-22, the threshold is the pivot itself)
    PivotMaxSim ms;
    ms.read_only = 1;
    const int st = launch_pivot_route(const_cast<float*>(scores), nq, n, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, bmax, cnt, nq + 16, bound, thr,
                                      cap, ids, vals, cnt, flag, s, nullptr, nullptr, nullptr, &ms, scores, ld);
    if (st != RL_OK) return st;
    RL_TRY(launch_merge_topk(vals, ids, 1, nq, cap, k, out_scores, out_ids, s, cnt));
    // more than 4096 scores reach the pivot (massive ties), or fewer than k groups have a usable maximum (NaN / -inf nearly everywhere): the
    // radix selection answers, behind the flag
    return launch_topk(scores, nq, n, ld, k, ws, out_scores, out_ids, s, flag);
}

static int hist_grid(int64_t n, int32_t nq) {
    return (int)std::max<int64_t>(1, std::min<int64_t>((n + 4095) / 4096, nq >= 64 ? 64 : 512));
}

int launch_transform_hist(float* scores, int32_t nb, int64_t n, int64_t ld, const float* row_norm, const float* row_sumsq,
                          const float* queries, int32_t dim, int mode, SelectWorkspace& ws, hipStream_t s, float pre_scale,
                          const uint32_t* run_if, uint32_t* zero_words, int n_zero, const HiBound* bound) {
    if (n <= 0 || nb <= 0) return RL_OK;
    const HiBound hb = bound ? *bound : HiBound{};
    if (n_zero < 0 || n_zero > 256 || (n_zero > 0 && !zero_words)) return RL_ERR_INVALID;
    RL_TRY(select_workspace_reserve(ws, nb, s));
    ws.dirty = true;  // the histogram rows stay non-zero until launch_topk(have_hist) has run its final kernel
    hipLaunchKernelGGL(transform_hist_kernel, dim3(hist_grid(n, nb), nb), dim3(256), 0, s, scores, n, ld, row_norm, row_sumsq,
                       queries, (int)dim, mode, ws.hist, pre_scale, run_if, zero_words, n_zero, hb);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_topk(const float* scores, int32_t nq, int64_t n, int64_t ld, int32_t k, SelectWorkspace& ws,
                float* out_scores, int32_t* out_ids, hipStream_t s, const uint32_t* run_if, bool have_hist, const HiEmit* emit) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (emit && (!emit->m || !emit->ids || !emit->cnt || !emit->flag || emit->cap < 1 || (emit->row_norm && !emit->norms))) return RL_ERR_INVALID;
    const HiEmit em = emit ? *emit : HiEmit{};
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "top-k: k must be <= 2048");
    if (n >= (int64_t)0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "top-k: more than 2^31-2 elements per query");
    if (have_hist) {
        if (nq > ws.capacity_queries || !ws.dirty) return fail(RL_ERR_INVALID, "top-k: no histogram in the workspace");
    } else {
        RL_TRY(select_workspace_reserve(ws, nq, s));
    }
    // The block route (round 6): a query whose scores one block can read twice out of L2 is selected by ONE launch that touches no workspace.
    // With the prefilter also BATCHES over more scores per query (up to 8 M: MaxSim chunk scores of a corpus of more than 262 144 chunks crowd
    // into one bin of the radix selection below -- its exact slow path, one block reading the scores three more times, was 1.1 ms of an 18.6 ms
    // step over 300 k chunks; a block per query reading them twice is 0.08)
    const bool big_batch = ws.block_route >= 2 && k <= PREFILTER_MAX_K && nq >= 16 && n <= (int64_t(8) << 20);
    if (!run_if && !have_hist && !emit && ws.block_route && n > 0 && (n <= BLOCK_ROUTE_MAX_N || big_batch)) {
        hipLaunchKernelGGL(topk_block_kernel, dim3(nq), dim3(1024), 0, s, scores, n, ld, k, out_scores, out_ids, ws.block_route >= 2 ? 1 : 0);
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    ws.dirty = true;
    // (Filter and final step as ONE launch by the last-block pattern measured slower at B = 1 over 1 M scores -- 0.628 / 0.679 ms per
    // query against 0.613 / 0.662 ms: the agent-scope release / acquire costs more than the launch it saves -- and was removed in round 3.)
    if (n > 0) {
        const int bx = hist_grid(n, nq);
        if (!have_hist) hipLaunchKernelGGL(topk_hist_kernel, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, ws.hist, run_if);
        hipLaunchKernelGGL(topk_filter_kernel, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, k, ws.hist, ws.sel, ws.cand, run_if, em.m, em.l2);
    }
    hipLaunchKernelGGL(topk_final_kernel, dim3(nq), dim3(1024), 0, s, scores, n, ld, k, ws.hist, ws.sel, ws.cand, out_scores, out_ids, run_if, em);
    RL_HIP(hipGetLastError());
    ws.dirty = false;  // every row this selection touched is zero again once the final kernel has run
    return RL_OK;
}

// ---- rank cut: the order-first-then-filter branch of the reference's vector search -----------------------------------------
// src/raglite/_search.py:120-141: when the metadata filter matches more than 100 000 rows the reference first cuts the table to
// its `LIMIT 1_000_000` nearest rows and filters THOSE.  Here: per query, the exact key T of the L-th best score by a
// three-level radix select (11 + 11 + 10 bits, the histogram machinery of the top-k above), ties on T taken in row order
// (SQL leaves them unspecified; the oracle and this file take the lowest rows) -- every row outside the L best, and every
// row whose chunk fails the filter, is set to -inf before the ordinary top-k.  5 passes over the [B x N] scores; the corpus
// scan that produced them read dim times as much.
constexpr int RANK_CHUNK = 4096;  // rows per block of the ordered passes

// Walks `levels` (1..3) histogram levels of query q (hq: [3][HIST_BINS], level 2 uses 1024 bins) from `need` = L: on
// return prefix = the key's leading 11 / 22 / 32 bits, need = elements still to take inside that prefix.  Every thread must
// call; h: HIST_BINS words of LDS.
__device__ __forceinline__ void rank_prefix(const uint32_t* __restrict__ hq, int levels, uint32_t L, uint32_t* h, uint32_t* scratch,
                                            uint32_t* thr, uint32_t& prefix, uint32_t& need) {
    prefix = 0;
    need = L;
    for (int lv = 0; lv < levels; ++lv) {
        const int nbins = lv == 2 ? 1024 : HIST_BINS;
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) h[i] = hq[lv * HIST_BINS + i];
        __syncthreads();
        find_threshold_bin(h, nbins, need, scratch, thr);
        prefix = (prefix << (lv == 2 ? 10 : 11)) | thr[0];
        need -= thr[1];
        __syncthreads();
    }
}

// LEVEL 0: histogram of the key's top 11 bits; LEVEL 1: of bits [20:10] among keys whose top 11 bits are the level-0
// threshold bin; LEVEL 2: of bits [9:0] among keys matching the 22-bit prefix.
template <int LEVEL>
__global__ __launch_bounds__(256) void rank_level_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, uint32_t L,
                                                          uint32_t* __restrict__ hists) {
    __shared__ uint32_t h[HIST_BINS];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t thr[2];
    const int q = blockIdx.y;
    uint32_t* hq = hists + (int64_t)q * 3 * HIST_BINS;
    uint32_t prefix, need;
    rank_prefix(hq, LEVEL, L, h, scratch, thr, prefix, need);
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const float* s = scores + (int64_t)q * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t key = score_key(s[i]);
        if (LEVEL == 0) atomicAdd(&h[key >> 21], 1u);
        else if (LEVEL == 1) { if ((key >> 21) == prefix) atomicAdd(&h[(key >> 10) & 2047u], 1u); }
        else { if ((key >> 10) == prefix) atomicAdd(&h[key & 1023u], 1u); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 256)
        if (h[i]) atomicAdd(&hq[LEVEL * HIST_BINS + i], h[i]);
}

// Keys equal to the threshold T in this block's RANK_CHUNK rows -> tie_counts[q][block].
__global__ __launch_bounds__(256) void rank_tie_count_kernel(const float* __restrict__ scores, int64_t n, int64_t ld, uint32_t L,
                                                              const uint32_t* __restrict__ hists, uint32_t* __restrict__ tie_counts) {
    __shared__ uint32_t h[HIST_BINS];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t thr[2];
    const int q = blockIdx.y;
    uint32_t T, need_eq;
    rank_prefix(hists + (int64_t)q * 3 * HIST_BINS, 3, L, h, scratch, thr, T, need_eq);
    const float* s = scores + (int64_t)q * ld;
    const int64_t base = (int64_t)blockIdx.x * RANK_CHUNK;
    uint32_t c = 0;
    for (int it = 0; it < RANK_CHUNK / 256; ++it) {
        const int64_t i = base + it * 256 + threadIdx.x;
        if (i < n) c += score_key(s[i]) == T;
    }
    uint32_t total;
    (void)block_inclusive_scan(c, scratch, total);
    if (threadIdx.x == 0) tie_counts[(int64_t)q * gridDim.x + blockIdx.x] = total;
}

// scores[i] = -inf unless row i is among the L best of its query (ties on T: the need_eq lowest rows) AND, with keep_bits,
// its bit is set (the metadata filter expanded to rows, tombstones included).
__global__ __launch_bounds__(256) void rank_cut_kernel(float* __restrict__ scores, int64_t n, int64_t ld, uint32_t L,
                                                        const uint32_t* __restrict__ hists, const uint32_t* __restrict__ tie_counts,
                                                        const uint32_t* __restrict__ keep_bits, const uint32_t* __restrict__ tie_base) {
    __shared__ uint32_t h[HIST_BINS];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t thr[2];
    const int q = blockIdx.y;
    uint32_t T, need_eq;
    rank_prefix(hists + (int64_t)q * 3 * HIST_BINS, 3, L, h, scratch, thr, T, need_eq);
    uint32_t before = (tie_base && threadIdx.x == 0) ? tie_base[q] : 0u;  // ties in the shards before this one (sharded cut), then in the blocks before this one
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += tie_counts[(int64_t)q * gridDim.x + b];
    uint32_t running;
    (void)block_inclusive_scan(before, scratch, running);
    __syncthreads();
    float* s = scores + (int64_t)q * ld;
    const int64_t base = (int64_t)blockIdx.x * RANK_CHUNK;
    for (int it = 0; it < RANK_CHUNK / 256; ++it) {
        const int64_t i = base + it * 256 + threadIdx.x;
        const uint32_t key = i < n ? score_key(s[i]) : 0u;
        const bool eq = i < n && key == T;
        uint32_t tot;
        const uint32_t incl = block_inclusive_scan(eq ? 1u : 0u, scratch, tot);
        bool keep = key > T || (eq && running + incl - 1 < need_eq);
        if (keep && keep_bits) keep = (keep_bits[i >> 5] >> (i & 31)) & 1u;
        if (i < n && !keep) s[i] = -INFINITY;
        running += tot;
        __syncthreads();  // scratch is reused by the next scan
    }
}

size_t rank_cut_scratch_bytes(int32_t nq, int64_t n) {
    return (size_t)nq * (3 * HIST_BINS + (size_t)((n + RANK_CHUNK - 1) / RANK_CHUNK)) * sizeof(uint32_t);
}

// The cut + filter, in place on scores [nq x ld] (elements that do not take part already -inf).  rank_limit >= n: only the
// filter.  scratch: rank_cut_scratch_bytes(nq, n).
int launch_rank_cut(float* scores, int32_t nq, int64_t n, int64_t ld, int64_t rank_limit, const uint32_t* keep_bits, void* scratch,
                    hipStream_t s) {
    if (nq <= 0 || n <= 0) return RL_OK;
    if (rank_limit < 1) return fail(RL_ERR_INVALID, "rank cut: rank_limit must be >= 1");
    if (rank_limit >= n) return keep_bits ? launch_mask_scores(scores, nq, n, ld, keep_bits, s) : RL_OK;
    if (n >= (int64_t)0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "rank cut: more than 2^31-2 elements per query");
    uint32_t* hists = static_cast<uint32_t*>(scratch);
    uint32_t* ties = hists + (size_t)nq * 3 * HIST_BINS;
    const int nblk = (int)((n + RANK_CHUNK - 1) / RANK_CHUNK);
    const uint32_t L = (uint32_t)rank_limit;
    RL_HIP(hipMemsetAsync(hists, 0, (size_t)nq * 3 * HIST_BINS * sizeof(uint32_t), s));
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((n + 4095) / 4096, nq >= 64 ? 64 : 512));
    hipLaunchKernelGGL(rank_level_kernel<0>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
    hipLaunchKernelGGL(rank_level_kernel<1>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
    hipLaunchKernelGGL(rank_level_kernel<2>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
    hipLaunchKernelGGL(rank_tie_count_kernel, dim3(nblk, nq), dim3(256), 0, s, scores, n, ld, L, hists, ties);
    hipLaunchKernelGGL(rank_cut_kernel, dim3(nblk, nq), dim3(256), 0, s, scores, n, ld, L, hists, ties, keep_bits, nullptr);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- the same cut in stages, for a corpus SHARDED over several indexes: the histograms of a level are additive,
// so a caller that sums each level's histogram over the shards between the stages makes every shard walk to the GLOBAL threshold key;
// ties on it are taken in global row order (tie_base = ties in the shards holding lower rows).  scratch as in launch_rank_cut, +
// nq x (HIST_BINS + 1) words behind it (a contiguous copy of one level, the per-query tie totals).
size_t rank_stage_scratch_bytes(int32_t nq, int64_t n) { return rank_cut_scratch_bytes(nq, n) + (size_t)nq * (HIST_BINS + 1) * sizeof(uint32_t); }

__global__ __launch_bounds__(256) void rank_level_copy_kernel(uint32_t* __restrict__ hists, int level, uint32_t* __restrict__ buf, int to_buf) {
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < HIST_BINS; i += 256) {
        uint32_t* a = hists + ((int64_t)q * 3 + level) * HIST_BINS + i;
        uint32_t* b = buf + (int64_t)q * HIST_BINS + i;
        if (to_buf) *b = *a; else *a = *b;
    }
}
__global__ __launch_bounds__(256) void rank_tie_total_kernel(const uint32_t* __restrict__ ties, int nblk, uint32_t* __restrict__ totals) {
    __shared__ uint32_t scratch[8];
    const int q = blockIdx.x;
    uint32_t c = 0;
    for (int b = threadIdx.x; b < nblk; b += 256) c += ties[(int64_t)q * nblk + b];
    uint32_t total;
    (void)block_inclusive_scan(c, scratch, total);
    if (threadIdx.x == 0) totals[q] = total;
}

// level 0 .. 2 of the radix walk over this shard's scores; the level's histogram [nq x HIST_BINS] -> level_out (device)
int launch_rank_stage_level(const float* scores, int32_t nq, int64_t n, int64_t ld, int64_t rank_limit, int level, void* scratch,
                            uint32_t* level_out, hipStream_t s) {
    if (nq <= 0 || level < 0 || level > 2 || rank_limit < 1) return fail(RL_ERR_INVALID, "rank cut: bad stage arguments");
    if (n >= (int64_t)0x7fffffff) return fail(RL_ERR_UNSUPPORTED, "rank cut: more than 2^31-2 elements per query");
    uint32_t* hists = static_cast<uint32_t*>(scratch);
    const uint32_t L = (uint32_t)std::min<int64_t>(rank_limit, 0xffffffffll);
    if (level == 0) RL_HIP(hipMemsetAsync(hists, 0, (size_t)nq * 3 * HIST_BINS * sizeof(uint32_t), s));
    if (n > 0) {
        const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((n + 4095) / 4096, nq >= 64 ? 64 : 512));
        if (level == 0) hipLaunchKernelGGL(rank_level_kernel<0>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
        else if (level == 1) hipLaunchKernelGGL(rank_level_kernel<1>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
        else hipLaunchKernelGGL(rank_level_kernel<2>, dim3(bx, nq), dim3(256), 0, s, scores, n, ld, L, hists);
    }
    hipLaunchKernelGGL(rank_level_copy_kernel, dim3(nq), dim3(256), 0, s, hists, level, level_out, 1);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// the level's histogram summed over the shards <- level_in (device)
int launch_rank_stage_set_level(int32_t nq, int level, void* scratch, const uint32_t* level_in, hipStream_t s) {
    if (nq <= 0 || level < 0 || level > 2) return fail(RL_ERR_INVALID, "rank cut: bad stage arguments");
    hipLaunchKernelGGL(rank_level_copy_kernel, dim3(nq), dim3(256), 0, s, static_cast<uint32_t*>(scratch), level, const_cast<uint32_t*>(level_in), 0);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// rows of this shard whose key is the global threshold key, per query -> totals_out [nq] (device)
int launch_rank_stage_ties(const float* scores, int32_t nq, int64_t n, int64_t ld, int64_t rank_limit, void* scratch, uint32_t* totals_out,
                           hipStream_t s) {
    if (nq <= 0) return RL_OK;
    uint32_t* hists = static_cast<uint32_t*>(scratch);
    uint32_t* ties = hists + (size_t)nq * 3 * HIST_BINS;
    const int nblk = (int)std::max<int64_t>(1, (n + RANK_CHUNK - 1) / RANK_CHUNK);
    const uint32_t L = (uint32_t)std::min<int64_t>(rank_limit, 0xffffffffll);
    hipLaunchKernelGGL(rank_tie_count_kernel, dim3(nblk, nq), dim3(256), 0, s, scores, n, ld, L, hists, ties);
    hipLaunchKernelGGL(rank_tie_total_kernel, dim3(nq), dim3(256), 0, s, ties, nblk, totals_out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}
// the cut (+ filter) in place, ties in row order after the tie_base[q] ties of the shards before this one
int launch_rank_stage_apply(float* scores, int32_t nq, int64_t n, int64_t ld, int64_t rank_limit, const uint32_t* keep_bits, void* scratch,
                            const uint32_t* tie_base, hipStream_t s) {
    if (nq <= 0 || n <= 0) return RL_OK;
    uint32_t* hists = static_cast<uint32_t*>(scratch);
    uint32_t* ties = hists + (size_t)nq * 3 * HIST_BINS;
    const int nblk = (int)((n + RANK_CHUNK - 1) / RANK_CHUNK);
    const uint32_t L = (uint32_t)std::min<int64_t>(rank_limit, 0xffffffffll);
    hipLaunchKernelGGL(rank_cut_kernel, dim3(nblk, nq), dim3(256), 0, s, scores, n, ld, L, hists, ties, keep_bits, tie_base);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- a8: max(sim) GROUP BY chunk ORDER BY max DESC LIMIT k  (src/raglite/_search.py:143-149) -----------
// Input: the a7 hits of each query, already sorted by (score desc, row asc).  A hit is kept iff no
// earlier hit belongs to the same chunk -- the first hit of a chunk carries the chunk's max, and the
// kept hits stay in (max desc, chunk ordinal asc) order because rows of a lower chunk ordinal are lower.
__global__ __launch_bounds__(256) void group_chunk_max_kernel(const float* __restrict__ hit_scores,
                                                               const int32_t* __restrict__ hit_rows,
                                                               int32_t num_hits,
                                                               const int64_t* __restrict__ chunk_offsets,
                                                               int64_t n_chunks, int32_t k,
                                                               float* __restrict__ out_scores,
                                                               int32_t* __restrict__ out_chunks,
                                                               int32_t* __restrict__ out_counts) {
    __shared__ int32_t chunk[K_MAX];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t running;
    const int q = blockIdx.x;
    const float* hs = hit_scores + (int64_t)q * num_hits;
    const int32_t* hr = hit_rows + (int64_t)q * num_hits;
    for (int i = threadIdx.x; i < num_hits; i += 256) {
        const int32_t r = hr[i];
        int32_t c = -1;
        if (r >= 0) {
            if (chunk_offsets) {  // largest c with offsets[c] <= r
                int64_t lo = 0, hi = n_chunks;  // invariant: offsets[lo] <= r < offsets[hi]
                while (hi - lo > 1) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (chunk_offsets[mid] <= (int64_t)r) lo = mid; else hi = mid;
                }
                c = (int32_t)lo;
            } else {
                c = r;
            }
        }
        chunk[i] = c;
    }
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < num_hits; base += 256) {
        const int i = base + threadIdx.x;
        bool keep = false;
        int32_t c = -1;
        if (i < num_hits) {
            c = chunk[i];
            keep = c >= 0;
            for (int j = 0; keep && j < i; ++j) keep = chunk[j] != c;
        }
        uint32_t tot;
        const uint32_t incl = block_inclusive_scan(keep ? 1u : 0u, scratch, tot);
        const uint32_t pos = running + incl - 1;
        __syncthreads();
        if (keep && pos < (uint32_t)k) {
            out_scores[(int64_t)q * k + pos] = hs[i];
            out_chunks[(int64_t)q * k + pos] = c;
        }
        if (threadIdx.x == 0) running += tot;
        __syncthreads();
    }
    const uint32_t count = running < (uint32_t)k ? running : (uint32_t)k;
    for (int i = count + threadIdx.x; i < k; i += 256) {
        out_scores[(int64_t)q * k + i] = -INFINITY;
        out_chunks[(int64_t)q * k + i] = -1;
    }
    if (threadIdx.x == 0) out_counts[q] = (int32_t)count;
}

int launch_group_chunk_max(const float* hit_scores, const int32_t* hit_rows, int32_t nq, int32_t num_hits,
                           const int64_t* chunk_offsets, int64_t n_chunks, int32_t k, float* out_scores,
                           int32_t* out_chunks, int32_t* out_counts, hipStream_t s) {
    if (nq <= 0) return RL_OK;
    if (num_hits > K_MAX) return fail(RL_ERR_UNSUPPORTED, "group-by-chunk: num_hits must be <= 2048");
    hipLaunchKernelGGL(group_chunk_max_kernel, dim3(nq), dim3(256), 0, s, hit_scores, hit_rows, num_hits,
                       chunk_offsets, n_chunks, k, out_scores, out_chunks, out_counts);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- section 8e: merge of per-shard top-k lists ---------------------------------------------------------
__global__ __launch_bounds__(1024) void merge_topk_kernel(const float* __restrict__ in_scores,
                                                           const int32_t* __restrict__ in_ids, int32_t n_lists,
                                                           int32_t n_queries, int32_t k_in, int32_t k,
                                                           float* __restrict__ out_scores,
                                                           int32_t* __restrict__ out_ids,
                                                           const uint32_t* __restrict__ counts, MergeTransform tr) {
    __shared__ uint64_t buf[MERGE_CAP];
    __shared__ float part[4];
    const int q = blockIdx.x;
    float qss = 0.f, qn = 0.f;
    if (tr.queries) {  // |q|^2 exactly as transform_kernel sums it: 256 lanes stride the query, wave sums, (p0 + p1) + (p2 + p3)
        if (threadIdx.x < 256) {
            float ss = 0.f;
            for (int c = threadIdx.x; c < tr.dim; c += 256) {
                const float v = tr.queries[(int64_t)q * tr.dim + c];
                ss = fmaf(v, v, ss);
            }
            ss = wave_sum(ss);
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
        }
        __syncthreads();
        qss = (part[0] + part[1]) + (part[2] + part[3]);
        qn = sqrtf(qss);
    }
    // counts (one list per query only): the list of query q holds min(counts[q], k_in) records, the rest of its k_in slots
    // was never written -- the sort then runs over the next power of two of THAT (the fused top-k's lists are a third full)
    const int total = counts ? (int)min(counts[q], (uint32_t)k_in) : n_lists * k_in;
    int p2 = 64;
    while (p2 < total) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += blockDim.x) {
        uint64_t key = 0ull;
        if (i < total) {
            const int l = i / k_in, j = i % k_in;
            const int64_t src = ((int64_t)l * n_queries + q) * k_in + j;
            const int32_t id = in_ids[src];
            if (id >= 0) {
                float v = in_scores[src];
                if (tr.queries) v = transform_score(v * 1.0f, tr.mode, tr.mode == SCAN_COSINE ? tr.row_norm[src] : 1.f, 0.f, qn, qss);
                key = make_key64(v, (uint32_t)id);
            }
        }
        buf[i] = key;
    }
    __syncthreads();
    if (counts && total <= 512) {
        // a candidate list of a bound-filtered search (one list, a few hundred DISTINCT ids at most): every record ranked by counting the larger
        // ones, a lane quad per record -- three barriers instead of the 28-36 of the sorting network (9-11 us per launch in every pipeline that
        // ends here; padding records -- key 0 -- are not ranked; 257..512 records: a lane PAIR per record)
        uint64_t* const fin = buf + 512;
        const int kk = total < k ? total : k;
        for (int i = threadIdx.x; i < kk; i += blockDim.x) fin[i] = 0ull;
        __syncthreads();
        const int per = total <= 256 ? 4 : 2;  // lanes per record
        const int a = threadIdx.x / per, sub = threadIdx.x % per;
        const uint64_t mine = a < total ? buf[a] : 0ull;
        uint32_t rank = 0;
        for (int j = sub; j < total; j += per) rank += buf[j] > mine;
        rank += __shfl_xor(rank, 1, 64);
        if (per == 4) rank += __shfl_xor(rank, 2, 64);
        if (sub == 0 && mine != 0ull && rank < (uint32_t)kk) fin[rank] = mine;
        __syncthreads();
        write_results(fin, kk, k, out_scores + (int64_t)q * k, out_ids + (int64_t)q * k);
        return;
    }
    bitonic_sort_desc(buf, p2);
    write_results(buf, total < k ? total : k, k, out_scores + (int64_t)q * k, out_ids + (int64_t)q * k);
}

// Half-bytes batched search (api.hip: search_rows_fused_hi, experimental): query q's candidate list holds min(counts[q], k_in)
// records (APPROXIMATE score, row).  Sorted by (score desc, row asc), the records whose score reaches (k-th best) - window[q]
// form a prefix -- every row that can be in the exact top-k when |approximate - exact| <= window[q] / 2 -- and their rows go to
// out_ids[q * cap2 ..], out_cnt[q] of them (any order is fine for the caller: it re-scores and ranks them).  More than cap2
// sets *flag.  Fewer than k records: all of them.
__global__ __launch_bounds__(1024) void list_prefix_kernel(const float* __restrict__ in_scores, const int32_t* __restrict__ in_ids,
                                                            int32_t k_in, int32_t k, const uint32_t* __restrict__ counts,
                                                            const float* __restrict__ window, int32_t cap2, int32_t* __restrict__ out_ids,
                                                            uint32_t* __restrict__ out_cnt, uint32_t* __restrict__ flag) {
    __shared__ uint64_t buf[MERGE_CAP];
    __shared__ uint32_t n_pass;
    const int q = blockIdx.x;
    const int total = (int)min(counts[q], (uint32_t)k_in);
    int p2 = 64;
    while (p2 < total) p2 <<= 1;
    for (int i = threadIdx.x; i < p2; i += blockDim.x) {
        uint64_t key = 0ull;
        if (i < total) {
            const int64_t src = (int64_t)q * k_in + i;
            const int32_t id = in_ids[src];
            if (id >= 0) key = make_key64(in_scores[src], (uint32_t)id);
        }
        buf[i] = key;
    }
    if (threadIdx.x == 0) n_pass = 0u;
    __syncthreads();
    bitonic_sort_desc(buf, p2);  // (ends with a barrier)
    const uint64_t kth = (k <= p2) ? buf[k - 1] : 0ull;
    const uint32_t k32 = (uint32_t)(kth >> 32);
    const float thr = (kth != 0ull && k32 != 0u) ? key_score(k32) - window[q] : -INFINITY;  // (NaN window -> nothing passes -> flag below)
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const uint64_t key = buf[i];
        const uint32_t s32 = (uint32_t)(key >> 32);
        if (key != 0ull && s32 != 0u && key_score(s32) >= thr) {
            ++mine;
            if (i < cap2) out_ids[(int64_t)q * cap2 + i] = (int32_t)(0xffffffffu - (uint32_t)key);  // sorted: the passing records are a prefix
        }
    }
    if (mine) atomicAdd(&n_pass, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = n_pass;
        out_cnt[q] = n < (uint32_t)cap2 ? n : (uint32_t)cap2;
        if (n > (uint32_t)cap2 || !(window[q] >= 0.f) || (total > 0 && n == 0u)) atomicOr(flag, 1u);
    }
}

int launch_list_prefix(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                       const float* window, int32_t cap2, int32_t* out_ids, uint32_t* out_cnt, uint32_t* flag, hipStream_t s) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (k_in > MERGE_CAP || cap2 < 1) return fail(RL_ERR_UNSUPPORTED, "list_prefix: k_in must be <= 8192");
    hipLaunchKernelGGL(list_prefix_kernel, dim3(nq), dim3(1024), 0, s, in_scores, in_ids, k_in, k, counts, window, cap2, out_ids, out_cnt, flag);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- round 5: the two list steps of the fused top-k by SELECTION instead of a sort ------------------------------------------------------
// list_prefix_kernel and the merge between the two rounds of the candidate pass sort a whole list (bitonic over up to 8192 64-bit keys: 66
// barrier-separated stages at 2048) to learn ONE number -- the k-th best score of the list -- and then, respectively, keep what lies within a
// window of it / raise a threshold to it.  A radix select on the 32-bit order-preserving score keys finds that number in four passes of
// 256 bins (three barriers each), every thread holding its share of the list in registers.  Ties on the k-th score do not matter to either
// caller: both compare SCORES with (k-th score - window).
//   MODE_RAISE == false (list_prefix): the rows whose score reaches (k-th best) - window[q] go to out_ids[q * cap2 ..] in any order (the
//     caller re-scores and ranks them), out_cnt[q] of them; more than cap2, a NaN window or an empty result set *flag -- list_prefix_kernel's
//     contract (fewer than k records: all of them).
//   MODE_RAISE == true: thr[q] = max(thr[q], (k-th best) - window[q]) (fewer than k records: unchanged) -- launch_merge_topk +
//     launch_raise_threshold in one launch.
template <bool MODE_RAISE>
__global__ __launch_bounds__(1024) void list_select_kernel(const float* __restrict__ in_scores, const int32_t* __restrict__ in_ids, int32_t k_in,
                                                            int32_t k, const uint32_t* __restrict__ counts, const float* __restrict__ window,
                                                            int32_t cap2, int32_t* __restrict__ out_ids, uint32_t* __restrict__ out_cnt,
                                                            uint32_t* __restrict__ flag, float* __restrict__ thr) {
    constexpr int PER = MERGE_CAP / 1024;  // records per thread (8)
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_prefix, sh_need, n_pass;
    const int q = blockIdx.x;
    const int total = (int)min(counts[q], (uint32_t)k_in);
    uint32_t key[PER];
    int32_t id[PER];
    uint32_t valid = 0u;  // records of the list that can rank (a real row, a comparable score)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = (int)threadIdx.x + j * 1024;
        key[j] = 0u;
        id[j] = -1;
        if (i < total) {
            const int64_t src = (int64_t)q * k_in + i;
            id[j] = in_ids[src];
            if (id[j] >= 0) key[j] = score_key(in_scores[src]);
            valid += key[j] != 0u;
        }
    }
    if (threadIdx.x == 0) { sh_prefix = 0u; sh_need = (uint32_t)k; n_pass = 0u; }
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    if (valid) atomicAdd(&n_pass, valid);
    __syncthreads();
    const uint32_t n_valid = n_pass;
    __syncthreads();
    if (threadIdx.x == 0) n_pass = 0u;
    __syncthreads();
    float kth = -INFINITY;
    bool have_kth = false;
    if (n_valid >= (uint32_t)k) {  // (block-uniform)
        // the k-th largest key: digit by digit from the top
        for (int shift = 24; shift >= 0; shift -= 8) {
            const uint32_t prefix = sh_prefix;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if (key[j] != 0u && (shift == 24 || (key[j] >> (shift + 8)) == prefix)) atomicAdd(&hist[(key[j] >> shift) & 255u], 1u);
            __syncthreads();
            if (threadIdx.x < 64) {  // one wave walks the 256 bins from the top: lane l owns bins 4 l .. 4 l + 3
                const int l = (int)threadIdx.x;
                const uint32_t c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
                uint32_t above = c0 + c1 + c2 + c3;  // -> records in bins above this lane's four (suffix sum over the lanes, exclusive)
                uint32_t incl = above;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_down(incl, o, 64);
                    if (l + o < 64) incl += t;
                }
                above = incl - above;
                const uint32_t need = sh_need;
                // the bin b with  count(bins > b) < need <= count(bins >= b)
                const uint32_t a3 = above, a2 = a3 + c3, a1 = a2 + c2, a0 = a1 + c1;
                int b = -1;
                uint32_t ab = 0u;
                if (a3 < need && need <= a3 + c3) { b = 4 * l + 3; ab = a3; }
                else if (a2 < need && need <= a2 + c2) { b = 4 * l + 2; ab = a2; }
                else if (a1 < need && need <= a1 + c1) { b = 4 * l + 1; ab = a1; }
                else if (a0 < need && need <= a0 + c0) { b = 4 * l; ab = a0; }
                if (b >= 0) { sh_prefix = (prefix << 8) | (uint32_t)b; sh_need = need - ab; }
            }
            __syncthreads();
            if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
            __syncthreads();
        }
        kth = key_score(sh_prefix);
        have_kth = true;
    }
    if constexpr (MODE_RAISE) {
        if (threadIdx.x == 0 && have_kth) {
            const float t = kth - window[q];
            if (t > thr[q]) thr[q] = t;  // (NaN never raises it)
        }
    } else {
        const float t = have_kth ? kth - window[q] : -INFINITY;  // (NaN window -> nothing passes -> flag below)
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (key[j] != 0u && key_score(key[j]) >= t) {
                const uint32_t slot = atomicAdd(&n_pass, 1u);
                if (slot < (uint32_t)cap2) out_ids[(int64_t)q * cap2 + slot] = id[j];
            }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t n = n_pass;
            out_cnt[q] = n < (uint32_t)cap2 ? n : (uint32_t)cap2;
            if (n > (uint32_t)cap2 || !(window[q] >= 0.f) || (total > 0 && n == 0u)) atomicOr(flag, 1u);
        }
    }
}

int launch_list_select(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                       const float* window, int32_t cap2, int32_t* out_ids, uint32_t* out_cnt, uint32_t* flag, hipStream_t s) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (k_in > MERGE_CAP || cap2 < 1) return fail(RL_ERR_UNSUPPORTED, "list_select: k_in must be <= 8192");
    hipLaunchKernelGGL(list_select_kernel<false>, dim3(nq), dim3(1024), 0, s, in_scores, in_ids, k_in, k, counts, window, cap2, out_ids, out_cnt, flag,
                       static_cast<float*>(nullptr));
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// thr[q] = max(thr[q], (k-th best score of list q) - window[q]); lists with fewer than k records leave thr[q] alone
int launch_list_raise_threshold(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                                const float* window, float* thr, hipStream_t s) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (k_in > MERGE_CAP) return fail(RL_ERR_UNSUPPORTED, "list_raise_threshold: k_in must be <= 8192");
    hipLaunchKernelGGL(list_select_kernel<true>, dim3(nq), dim3(1024), 0, s, in_scores, in_ids, k_in, k, counts, window, 1, static_cast<int32_t*>(nullptr),
                       static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), thr);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

__global__ __launch_bounds__(256) void raise_threshold_kernel(float* __restrict__ thr, const float* __restrict__ kth, int32_t nq, int32_t k,
                                                               const float* __restrict__ window) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const float a = kth[(int64_t)q * k + (k - 1)];  // -inf: the subset held fewer than k rows that reached the first threshold
    const float t = a - window[q];
    if (t > thr[q]) thr[q] = t;  // (NaN never raises it)
}

int launch_raise_threshold(float* thr, const float* kth, int32_t nq, int32_t k, const float* window, hipStream_t s) {
    if (nq <= 0) return RL_OK;
    hipLaunchKernelGGL(raise_threshold_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, thr, kth, nq, k, window);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_merge_topk(const float* in_scores, const int32_t* in_ids, int32_t n_lists, int32_t nq, int32_t k_in,
                      int32_t k, float* out_scores, int32_t* out_ids, hipStream_t s, const uint32_t* counts, const MergeTransform* transform) {
    if (nq <= 0 || k <= 0) return RL_OK;
    if (transform && (n_lists != 1 || transform->mode == SCAN_L2 || (transform->mode == SCAN_COSINE && !transform->row_norm)))
        return fail(RL_ERR_INVALID, "merge: the fused transform needs n_lists == 1 and cosine (with norms) or dot");
    const MergeTransform tr = transform ? *transform : MergeTransform{};
    if ((int64_t)n_lists * k_in > MERGE_CAP) return fail(RL_ERR_UNSUPPORTED, "merge: n_lists * k_in must be <= 8192");
    if (counts && n_lists != 1) return fail(RL_ERR_INVALID, "merge: per-query counts need n_lists == 1");
    hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(1024), 0, s, in_scores, in_ids, n_lists, nq, k_in, k,
                       out_scores, out_ids, counts, tr);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
