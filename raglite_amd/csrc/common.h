// Shared declarations for libraglite_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "raglite_hip.h"

namespace rl {

// ---- host-side error plumbing ---------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define RL_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rl::fail(_e == hipErrorOutOfMemory ? RL_ERR_NOMEM : RL_ERR_HIP,         \
                              std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

#define RL_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != RL_OK) return _s; \
    } while (0)

// Experiment switches (kernel A/B variants, timing skeletons, s_memtime traces) are read from the environment ONLY in experiment
// builds (-DRAGLITE_EXPERIMENTS: raglite_amd._build.build(experiments=True) -> libraglite_hip_exp.so, used by scripts/gpu_calls/).  The
// shipped library reads no environment variable: there every switch folds to "off" at compile time.
#ifdef RAGLITE_EXPERIMENTS
inline const char* exp_env(const char* name) { return std::getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

constexpr int WAVE = 64;
constexpr int K_MAX = 2048;       // largest top-k the selection stage supports
constexpr int HIST_BINS = 2048;   // top 11 bits of the orderable score key
constexpr int CAND_CAP = 4096;    // threshold-bin candidates kept per query before the slow path
constexpr int MERGE_CAP = 8192;   // n_lists * k_in accepted by rl_merge_topk

// ---- device helpers ----------------------------------------------------------------------------
#ifdef __HIPCC__

// Monotone map float -> uint32: larger float <=> larger key; every NaN -> 0 (ranks last).
__device__ __forceinline__ uint32_t score_key(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// 64-bit total order: (score desc, id asc)  <=>  key64 desc.  key64 == 0 is the padding value.
__device__ __forceinline__ uint64_t make_key64(float score, uint32_t id) {
    return ((uint64_t)score_key(score) << 32) | (uint64_t)(0xffffffffu - id);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}


// Wave-uniform id of the calling wave inside its block (provably uniform for the compiler).
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

#endif  // __HIPCC__

// Grid of a persistent streaming kernel: exactly one resident generation of workgroups (CUs x the
// occupancy the runtime reports for this kernel), capped by the work available.  Measured on the B = 1
// scan: 5 blocks/CU (= its occupancy) 0.71 ms, 8 blocks/CU 0.81 ms -- a second, partial generation of
// blocks costs a tail.
template <class Kernel>
inline int persistent_grid(Kernel kernel, int block_threads, int64_t max_useful_blocks) {
    static thread_local int cached_cu = 0;
    if (cached_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cached_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cached_cu <= 0)
            cached_cu = 256;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, 0) != hipSuccess || per_cu <= 0)
        per_cu = 4;
    return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)cached_cu * per_cu, max_useful_blocks));
}

// ---- kernel launchers (one per .hip file) -------------------------------------------------------
int launch_synth(float* dst, int64_t start, int64_t count, uint64_t seed, int kind, hipStream_t s);

int launch_pool_norm(const float* tokens, int32_t dim, const int64_t* span_begin, const int64_t* span_end,
                     int64_t n_spans, int normalize, double eps, float* out_f32, uint16_t* out_f16,
                     hipStream_t s);

// scan.hip
enum ScanMode { SCAN_RAW_DOT = 0, SCAN_COSINE = 1, SCAN_DOT = 2, SCAN_L2 = 3 };

// The metric transform of a raw dot product (transform_kernel in scan.hip and transform_hist_kernel in select.hip share
// these statements, so both give the same bits): qss = |q|^2 summed as block_query_sumsq does, qn = sqrtf(qss).
__device__ __forceinline__ float transform_score(float d, int mode, float row_norm, float row_sumsq, float qn, float qss) {
    if (mode == SCAN_COSINE) return 1.0f - (1.0f - d / (row_norm * qn));
    if (mode == SCAN_DOT) return 1.0f + d;
    if (mode == SCAN_L2) return 1.0f - sqrtf(fmaxf(row_sumsq + qss - 2.0f * d, 0.f));
    return d;
}
// scores[b * ld + row] for b < nb (nb <= 4 per launch handled inside), rows < n.
int launch_scan_rows(const float* E, int64_t n, int32_t dim, const float* queries, int32_t nb,
                     const float* row_norm, int mode, float* scores, int64_t ld, hipStream_t s, const uint32_t* run_if = nullptr);
// scan16.hip: the same over an fp16-stored corpus (dim <= 1024) -- or over the HI plane of a WIDE fp32 index (dim <= 4096, nb <= 4: raw dots)
int launch_scan_rows16(const uint16_t* E, int64_t n, int32_t dim, const float* queries, int32_t nb,
                       const float* row_norm, int mode, float* scores, int64_t ld, hipStream_t s);
int launch_row_norms16(const uint16_t* E, int64_t n, int32_t dim, float* norm, float* sumsq, hipStream_t s);
int launch_row_norms(const float* E, int64_t n, int32_t dim, float* norm, float* sumsq, hipStream_t s);
int launch_cast_f16(const float* src, uint16_t* dst, int64_t count, hipStream_t s);
// dst = fp16(src * scale) rounded to nearest even (the HI plane); count % 8 == 0, 16-byte aligned pointers
int launch_cast_f16_scaled(const float* src, uint16_t* dst, int64_t count, float scale, hipStream_t s);
int launch_fill_f32(float* dst, float value, int64_t count, hipStream_t s);
int launch_scale_f32(const float* src, float* dst, float factor, int64_t count, hipStream_t s, uint32_t* zero_words = nullptr,
                     int n_zero = 0);  // hi_filter.hip: dst = src * factor (and n_zero words set to zero on the way)
// in-place metric transform of raw dots: scores[b*ld+i] (i<n), per-row norm / sumsq, per-query norm.
// (pre_scale: the raw dots are multiplied by it first -- a power of two, exact; run_if as in launch_topk)
int launch_transform(float* scores, int32_t nb, int64_t n, int64_t ld, const float* row_norm,
                     const float* row_sumsq, const float* queries, int32_t dim, int mode, hipStream_t s, float pre_scale = 1.0f,
                     const uint32_t* run_if = nullptr);

// select.hip
struct SelectWorkspace {      // device buffers sized for `capacity_queries`
    uint32_t* hist = nullptr; // [B][HIST_BINS + 8]  (bins, then counters).  ALL ZERO between selections: zeroed when
                              // allocated, and the final kernel of every selection zeroes its query's row again once it
                              // has read it -- no memset launch in front of every top-k
    uint64_t* sel = nullptr;  // [B][K_MAX]
    uint64_t* cand = nullptr; // [B][CAND_CAP]
    uint32_t* pv = nullptr;   // [3 B + 16] counters, flag block, bounds and thresholds of launch_topk_pivot
    int32_t capacity_queries = 0;
    bool dirty = false;       // a launch sequence was cut short by an error: re-zero `hist` before the next use
    int block_route = 2;      // RL_OPT_TOPK_BLOCK: selections of <= 256 k scores per query in ONE launch, one block per query (select.hip); 2: with the thread-maximum prefilter
};
int select_workspace_reserve(SelectWorkspace& ws, int32_t n_queries, hipStream_t s);
void select_workspace_free(SelectWorkspace& ws);
// run_if != nullptr: the three kernels return at once unless *run_if != 0 (the guarded dense fallback of the fused top-k).
// have_hist: the histogram of `scores` is already in ws.hist (launch_transform_hist) -- skip that pass.
// emit (the half-bytes row search, api.hip: search_rows_hi): the selection also lists every row whose score is within 2 m[q] of the k-th best
// -- the candidates of the exact re-scoring -- without another pass over the scores (topk_filter_kernel / topk_final_body have the argument)
struct HiEmit {
    const float* m = nullptr;         // [n_queries] error bound per query (launch_transform_hist: HiBound)
    int32_t cap = 0;                  // slots per candidate list
    int32_t* ids = nullptr;           // [n_queries x cap] rows; nullptr = no emission
    float* norms = nullptr;           // [n_queries x cap] their norms (with row_norm)
    const float* row_norm = nullptr;  // [n] (cosine) or nullptr
    uint32_t* cnt = nullptr;          // [n_queries], zero on entry
    uint32_t* flag = nullptr;         // set on overflow / unusable threshold
    float* thr = nullptr;             // [n_queries] (optional) the thresholds, for diagnostics
    int l2 = 0;                       // the scores are l2 similarities 1 - sqrt(D), m the bound of D: lower_threshold's other form
};
int launch_topk(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int32_t k,
                SelectWorkspace& ws, float* out_scores, int32_t* out_ids, hipStream_t s, const uint32_t* run_if = nullptr,
                bool have_hist = false, const HiEmit* emit = nullptr);
// Transform + exact top-k of raw dots in ONE guarded launch, one block per query (slow: the fallback of the half-bytes row search)
int launch_guarded_select(float* scores, int32_t nb, int64_t n, int64_t ld, int32_t k, const float* row_norm, const float* row_sumsq,
                          const float* queries, int32_t dim, int mode, float pre_scale, float* out_scores, int32_t* out_ids,
                          const uint32_t* run_if, hipStream_t s, uint32_t* host_flag = nullptr);
// launch_transform (scan.hip) + the selection's histogram pass in ONE launch: raw dots -> similarities in place, and their
// 2048-bin key histogram into ws.hist (same statements as transform_kernel: same bits).  Follow with launch_topk(have_hist).
// pre_scale: the raw dots are multiplied by it first (a power of two: exact; 1 = the plain transform); run_if as in launch_topk.
// zero_words / n_zero (<= 256): words this launch also sets to zero (the candidate counters + flag of a B <= 16 row search: no memset launch).
// bound: m_out[b] = the half-bytes search's error bound of query b -- cosine: m_rel; dot: m_rel * e_norm_bound * |q_b| + 2^-22
// The half-bytes search's candidate threshold (hi_filter.hip has the derivation of the l2 form in front of approx_threshold_kernel)
__device__ __forceinline__ float l2_delta(float a, float eps, float e_max, float qn) {
    const float t = e_max + qn;
    return (2.0f * a * qn + 2.0f * eps * t * t) * 1.0001f;
}
// the candidate threshold under a lower bound P of the k-th best approximate similarity; m: the query's bound (l2: delta, in squared distance)
__device__ __forceinline__ float lower_threshold(float P, float m, bool l2) {
    if (!l2) return P - 2.0f * m;
    const float r = fmaxf(1.0f - P, 0.f);
    return 1.0f - sqrtf(fmaf(r, r, 2.0f * m)) * (1.0f + 0x1p-18f) - 0x1p-20f;
}

struct HiBound {
    float* m_out = nullptr;
    float m_rel = 0.f, e_norm_bound = 0.f;
    float e_max = 0.f;  // l2 (round 6): max |e|; m_rel = the rounding term per |q| |e| (sum_eps), e_norm_bound = the dot bound per |q| -- m_out[b] is then
                        // the bound of the SQUARED distance, hi_filter.hip: l2_delta
};
int launch_transform_hist(float* scores, int32_t nb, int64_t n, int64_t ld, const float* row_norm, const float* row_sumsq,
                          const float* queries, int32_t dim, int mode, SelectWorkspace& ws, hipStream_t s, float pre_scale = 1.0f,
                          const uint32_t* run_if = nullptr, uint32_t* zero_words = nullptr, int n_zero = 0, const HiBound* bound = nullptr);
int launch_group_chunk_max(const float* hit_scores, const int32_t* hit_rows, int32_t n_queries,
                           int32_t num_hits, const int64_t* chunk_offsets, int64_t n_chunks, int32_t k,
                           float* out_scores, int32_t* out_chunks, int32_t* out_counts, hipStream_t s);
// The metric transform of raw dots applied on the way into the merge (n_lists == 1): record j of query q is transform_score(in_scores, mode,
// row_norm[q * k_in + j], .., |queries[q]|) -- the statements of transform_kernel (same bits), without its launch.
struct MergeTransform {
    const float* row_norm = nullptr;  // [n_queries x k_in] (cosine), else unused
    const float* queries = nullptr;   // [n_queries x dim]; nullptr = no transform
    int dim = 0, mode = 0;
};
int launch_merge_topk(const float* in_scores, const int32_t* in_ids, int32_t n_lists, int32_t n_queries,
                      int32_t k_in, int32_t k, float* out_scores, int32_t* out_ids, hipStream_t s,
                      const uint32_t* counts = nullptr,  // counts (n_lists == 1): records list q really holds
                      const MergeTransform* transform = nullptr);

// select.hip, rank cut (order-first-then-filter, src/raglite/_search.py:120-141): rows outside the rank_limit best of their
// query, and rows whose keep bit is clear, become -inf
// the cut in stages (a corpus sharded over several indexes sums each level's histogram over the shards between them)
size_t rank_stage_scratch_bytes(int32_t n_queries, int64_t n);
int launch_rank_stage_level(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int64_t rank_limit, int level, void* scratch,
                            uint32_t* level_out, hipStream_t s);
int launch_rank_stage_set_level(int32_t n_queries, int level, void* scratch, const uint32_t* level_in, hipStream_t s);
int launch_rank_stage_ties(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int64_t rank_limit, void* scratch,
                           uint32_t* totals_out, hipStream_t s);
int launch_rank_stage_apply(float* scores, int32_t n_queries, int64_t n, int64_t ld, int64_t rank_limit, const uint32_t* keep_bits,
                            void* scratch, const uint32_t* tie_base, hipStream_t s);
size_t rank_cut_scratch_bytes(int32_t n_queries, int64_t n);
int launch_rank_cut(float* scores, int32_t n_queries, int64_t n, int64_t ld, int64_t rank_limit, const uint32_t* keep_bits,
                    void* scratch, hipStream_t s);

// hi_filter.hip: helpers of the half-bytes single-query search (api.hip: search_rows_hi)
int launch_approx_threshold(const float* topk, int32_t nb, int32_t k, const float* queries, int32_t dim, int mode, float m_rel,
                            float e_norm_bound, float* thr, uint32_t* cnt, uint32_t* flag, hipStream_t s, float e_max = 0.f);  // also zeroes cnt[0..nb) and *flag
// top_s / top_i [nb x k] (optional): the approximate top-k in selection order -- only entries ranking BELOW its k-th one are collected
struct PivotMaxSim {                  // the MaxSim flavour of the pivot route (hi_filter.hip: transform_bmax_kernel)
    int nq = 0;                       // query vectors per query (0: a row search)
    int64_t q_stride = 0;             // floats between two queries
    float m_abs = 0.f;                // m_b = m_abs * sum_i |q_i|
    int32_t* fill_ids = nullptr;      // [nb x cap] candidate lists to pre-fill with -1
    int64_t fill_n = 0;               // (set by the launcher)
    int per_wave = 0;                 // (set by the launcher)
    int read_only = 0;                // the scores are not rewritten (an identity transform: launch_topk_pivot)
};
size_t pivot_scratch_words(int32_t nb);
bool pivot_route_takes(int64_t n, int32_t k);
int launch_pivot_route(float* scores, int32_t nb, int64_t n, int64_t ld, int32_t k, const float* row_norm, const float* row_sumsq, const float* queries,
                       int32_t dim, int mode, float pre_scale, uint64_t* bmax, uint32_t* zero_words, int n_zero, const HiBound& bound, float* thr,
                       int32_t cap, int32_t* ids, float* norms, uint32_t* cnt, uint32_t* flag, hipStream_t s, const float* E = nullptr,
                       float* gather_out = nullptr, bool* gathered = nullptr, const PivotMaxSim* maxsim = nullptr, const float* aux_src = nullptr,
                       int64_t aux_ld = 0);
// Exact top-k of scores that CROWD (l2 similarities 1 - |e - q| of a big corpus share one exponent and a few mantissa bits: the radix selection's
// threshold bin then holds the whole corpus and its one-block slow path takes 2 ms per query): the pivot route's group maxima do not care how the
// scores are distributed.  k <= 128, >= 3 k groups, <= 240 queries; RL_ERR_UNSUPPORTED otherwise.  Same results as launch_topk.
int launch_topk_pivot(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int32_t k, SelectWorkspace& ws, float* out_scores,
                      int32_t* out_ids, hipStream_t s);
int launch_collect_above(const float* scores, int32_t nb, int64_t n, int64_t ld, const float* thr, const float* row_norm, int32_t cap,
                         int32_t* ids, float* norms, uint32_t* cnt, uint32_t* flag, hipStream_t s, const float* top_s = nullptr,
                         const int32_t* top_i = nullptr, int32_t k = 0);
// thr[b] = min_j exact[b][j] - m[b] (the k-th best EXACT score of the approximate top-k bounds the k-th best overall from below);
// ids[b][0 .. k) = top_i[b], es[b][0 .. k) = exact[b], cnt[b] = k; unusable -> *flag, thr = +inf, cnt = 0.  See exact_threshold_kernel.
// fp16 queries over an fp16-stored corpus: widen the queries (exact), certify per query that the one-product pass lost nothing and hand its
// top-k out as the result (hi_filter.hip)
int launch_widen_f16(const uint16_t* src, float* dst, int64_t count, hipStream_t s);
int launch_f16_exact_finish(const float* Q, int32_t nq, int32_t dim, int64_t q_stride, const float* q_unscale, const float* top_s,
                            const int32_t* top_i, int32_t n_queries, int32_t k, float* out_s, int32_t* out_i, uint32_t* cnt, uint32_t* flag,
                            hipStream_t s);
int launch_exact_threshold(const float* exact, const int32_t* top_i, int32_t n_queries, int32_t k, float* m, int32_t cap, float* thr,
                           uint32_t* cnt, int32_t* ids, float* es, uint32_t* flag, hipStream_t s, const float* qsum = nullptr, float m_abs = 0.f,
                           float e_norm_max = 0.f, bool with_lo = false, const float* top_s = nullptr);  // top_s: the approximate top-k scores (-inf: masked)
// MaxSim flavour of the threshold: thr[b] = topk[b * k + k - 1] - 2 * m_rel * e_max * sum_i |q_i|; zeroes cnt[b]; sets *flag when
// the k-th score is unusable.  One block per query.
// q_unscale != nullptr (one-product pass: only the queries' fp16 hi halves were multiplied): q_unscale[2 * b] = 2^(ex - 14) of
// query b (the meta words of launch_query_planes), and the threshold drops by another 2 * e_norm_max * sum_i |q_lo,i| with
// q_lo,i = q_i - fp16(q_i * scale) / scale, recomputed here with the statement query_planes_kernel uses.
int launch_maxsim_threshold(const float* topk, int32_t n_queries, int32_t k, const float* Q, int32_t nq, int32_t dim, int64_t q_stride,
                            float m_rel, float e_max, float* thr, uint32_t* cnt, uint32_t* flag, hipStream_t s,
                            const float* q_unscale = nullptr, float e_norm_max = 0.f, float* m_out = nullptr);  // m_out[b] = the bound m_b
// A MaxSim batch over a sharded corpus (api.hip: rl_maxsim_batch_begin / _finish): out[b] = top[b][0 .. k) ++ {m_b} with neg2m[b] = -2 m_b;
// thr[b] = (k-th best of all shards' lists [world][B][k + 1]) - max_r m_r - m_rank, cnt[b] = 0, *flag on an unusable threshold
int launch_pack_approx(const float* top, const float* neg2m, int32_t B, int32_t k, float* out, hipStream_t s);
int launch_global_threshold(const float* lists, int32_t world, int32_t B, int32_t k, int32_t rank, float* thr, uint32_t* cnt, uint32_t* flag,
                            hipStream_t s);
// bits[0..2] = max |e|, max |e_lo|, max |e_lo| / |e| over the rows (float bit patterns, nudged up by 1e-6; start them at the
// values so far), e_lo = what the fp16 HI halves at `scale` drop
int launch_max_row_norm(const float* E, int64_t n_rows, int32_t dim, float scale, uint32_t* bits, hipStream_t s);
int launch_max_row_norm16(const uint16_t* E, int64_t n_rows, int32_t dim, uint32_t* bits, hipStream_t s);  // fp16-stored rows: bits[0] only
int launch_diag_blocks(const float* src, int64_t ld, int32_t k2, int64_t count, float* dst, hipStream_t s);
// Batched half-bytes search (experimental, api.hip: search_rows_fused_hi) -- see the kernels' comments in hi_filter.hip / select.hip
int launch_row_threshold(const float* topk, int32_t nb, int32_t k, const float* Q, int32_t dim, int mode, const float* q_unscale, float lo_ratio,
                         float lo_norm, float e_norm, float* thr, float* window, uint32_t* cnt, uint32_t* cnt2, uint32_t* flag, hipStream_t s,
                         float* thr_copy = nullptr, float sum_eps = 0x1p-12f);
int launch_row_dots(const float* E, int32_t dim, const float* Q, int32_t nb, const int32_t* rows, const uint32_t* cnt, int32_t cap, int mode,
                    const float* row_norm, const float* q_sumsq, float* out, hipStream_t s);
int launch_list_prefix(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                       const float* window, int32_t cap2, int32_t* out_ids, uint32_t* out_cnt, uint32_t* flag, hipStream_t s);
// the same contract by a radix SELECT of the list's k-th best score instead of a sort (round 5; out_ids in any order), and the step between
// the two rounds of the candidate pass -- thr[q] = max(thr[q], (k-th best of list q) - window[q]) -- as one launch of the same kernel
int launch_list_select(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                       const float* window, int32_t cap2, int32_t* out_ids, uint32_t* out_cnt, uint32_t* flag, hipStream_t s);
int launch_list_raise_threshold(const float* in_scores, const int32_t* in_ids, int32_t nq, int32_t k_in, int32_t k, const uint32_t* counts,
                                const float* window, float* thr, hipStream_t s);

// mask.hip: validity bitsets (metadata filter pushed down to the device, tombstones of deleted chunks)
int launch_expand_chunk_bits(const uint32_t* chunk_bits, const int32_t* row_to_chunk, int64_t n_rows,
                             const uint32_t* and_rows, uint32_t* row_bits, hipStream_t s);
int launch_mask_scores(float* scores, int32_t nb, int64_t n, int64_t ld, const uint32_t* bits, hipStream_t s);
int launch_fix_masked(const float* scores, int32_t* ids, int64_t count, hipStream_t s);
int launch_popcount(const uint32_t* bits, int64_t n, unsigned long long* out_dev, hipStream_t s);

// adapter_fit.hip: device half of update_query_adapter (best row per (query, chunk), row gather)
int launch_chunk_best_rows(const void* E, bool f16, int32_t dim, const float* Q, const int64_t* offsets,
                           int64_t n_chunks, const int32_t* cand, int32_t n_cand, int64_t n_items, int32_t* out_rows,
                           hipStream_t s);
int launch_rescore_l2(const void* E, bool f16, int32_t dim, const float* Q, const int32_t* rows, const float* ranked,
                      int32_t k, int64_t n_items, float* out, hipStream_t s);
int launch_permute_rows(const int32_t* rows_in, const int32_t* pos, int32_t k, int64_t n_items, int32_t* rows_out,
                        hipStream_t s);
// counts / cap (optional): rows[] is nq lists of cap slots of which list q holds min(counts[q], cap) -- the other slots are skipped (their
// output rows keep whatever they held)
int launch_gather_rows(const void* E, bool f16, int32_t dim, int64_t n_rows, const int32_t* rows, int64_t n, float* out,
                       hipStream_t s, const uint32_t* counts = nullptr, int32_t cap = 0);
int launch_compact_rows(const void* src, int64_t row_bytes, const int64_t* old_row, int64_t n, void* dst, hipStream_t s);

// partition_sim.hip: semantic-chunking similarities (src/raglite/_split_chunks.py:54-72), batched over documents
size_t partition_sim_scratch_bytes(int64_t n, int64_t n_docs, int32_t dim);
int launch_partition_similarity(const float* X, int64_t n, int32_t dim, const int64_t* doc_off, int64_t n_docs,
                                const uint8_t* sel, float* out, void* scratch, hipStream_t s);

// maxsim*.hip
int launch_row_to_chunk(const int64_t* chunk_offsets, int64_t n_chunks, int64_t n_rows, int32_t* row_to_chunk,
                        hipStream_t s);
// Fast path: dim in {128,256,384,512,768,1024}, nq <= 32 (wave-specialised MFMA streaming kernel).  mode 0: chunk MaxSim scores
// out[n_chunks]; mode 1: raw row dots out[q * ld + row].  Returns RL_ERR_UNSUPPORTED outside the fast path.
// split_scale > 0: SPLIT arithmetic (fp32 operands as fp16 hi + lo pairs, three fp16 MFMAs, fp32 accumulation) with the
// corpus scaled by that power of two; 0: the exact fp32 MFMA chain.
int launch_maxsim_stream(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                         const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                         float* out, int64_t ld, int n_cu, hipStream_t s, float split_scale = 0.f,
                         const uint32_t* run_if = nullptr);  // run_if: the kernel returns at once unless *run_if != 0
struct StreamSecondJob {  // launch_maxsim_stream_two: what grid row 1 runs
    const float* D = nullptr;
    int64_t n_rows = 0;
    float* out = nullptr;
    int64_t ld = 0;
    const uint32_t* run_if = nullptr;  // the row returns at once unless *run_if != 0 (nullptr: always runs)
};
int launch_maxsim_stream_two(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq, const int32_t* row_to_chunk,
                             const int64_t* chunk_offsets, int64_t n_chunks, float* out, int64_t ld, int n_cu, hipStream_t s, float split_scale,
                             const StreamSecondJob& job2);
int launch_maxsim_stream_batch(const void* D, bool f16, int64_t n_rows, int32_t dim, const float* Q, int32_t nq, int64_t q_stride,
                               int32_t n_queries, const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, float* out,
                               int64_t out_stride, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if = nullptr);
size_t query_split_bytes(int32_t dim, int32_t n_queries);
int launch_query_split(const float* Q, int32_t dim, int32_t nq, int64_t q_stride, int32_t n_queries, void* buf, bool f16_corpus,
                       hipStream_t s);
int launch_maxsim_stream2(const void* D, bool f16, int64_t n_rows, int32_t dim, const void* split_buf, int32_t n_queries,
                          int32_t first, int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks,
                          float* out, int64_t out_stride, int n_cu, hipStream_t s, float split_scale);
// maxsim_gemm.hip: eight queries per corpus pass over the pre-split (fp16 hi | lo) corpus image; `half`: the one-plane image of
// an fp16-stored corpus (split_scale = 1)
size_t planes_bytes(int64_t rows, int32_t dim, bool half = false);
int launch_presplit_rows(const float* E, int64_t first_row, int64_t n_rows, int32_t dim, float scale, void* planes, hipStream_t s);
int launch_preformat_rows16(const uint16_t* E, int64_t first_row, int64_t n_rows, int32_t dim, void* planes, hipStream_t s);
int launch_presplit_hi_rows(const float* E, int64_t first_row, int64_t n_rows, int32_t dim, float scale, void* planes, hipStream_t s);
size_t chunk_ends_words(int64_t rows);
int launch_chunk_ends(const int32_t* row_to_chunk, int64_t n_rows, uint32_t* ends, hipStream_t s);
size_t query_planes_bytes(int32_t dim, int32_t n_queries);
int launch_query_planes(const float* Q, int32_t dim, int32_t nq, int64_t q_stride, int32_t n_queries, void* buf, hipStream_t s,
                        uint32_t* zero_words = nullptr, int n_zero = 0);
int launch_maxsim_gemm(const void* planes, int64_t n_rows, int32_t dim, const void* qbuf, int32_t n_queries, int32_t first,
                       int32_t n_q, int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, const uint32_t* ends_bits,
                       float* out, int64_t out_stride, int n_cu, hipStream_t s, float split_scale, bool half = false,
                       const uint32_t* run_if = nullptr, bool hi_only = false, bool all_passes = false);  // all_passes: n_q > 8 in one launch
// maxsim_pp.hip: the approximate pass of the headline pipeline -- SIXTEEN queries per pass over a one-plane image, one product
constexpr int32_t PP_PASS_QUERIES = 16;
int launch_maxsim_pp(const void* image, int64_t n_rows, int32_t dim, const void* qbuf, int32_t n_queries, int32_t first, int32_t n_q,
                     int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, const uint32_t* ends_bits, float* out,
                     int64_t out_stride, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if = nullptr);
// the matrix pipe's sustained fp16 rate under the pass kernel's own MFMA stream (no loads, no epilogue): rl_time_kernel kind 9
constexpr int32_t MFMA_RATE_ITERS = 1000;  // x 32 MFMAs per wave: the MFMA count of a wave over one pass of 1 M rows x 1024
int launch_mfma_f16_rate(float* out, int n_cu, int32_t iters, hipStream_t s, double* flops);
// the candidate pass of the fused row top-k on that kernel's tile (MODE 2; see maxsim_pp.hip) -- declared after CandArgs below
size_t score_planes_scratch_floats(int32_t nb, int32_t dim);
int launch_score_planes(const void* planes, int64_t n_rows, int32_t dim, const float* Q, int32_t nb, float* scores, int64_t ld,
                        const float* row_norm, const float* row_sumsq, float* scratch, int mode, int n_cu, hipStream_t s, float split_scale,
                        bool half = false);
struct CandArgs { const float* tau; int32_t tau_stride; float* scores; int32_t* ids; uint32_t* cnt; uint32_t* overflow; int32_t cap; };
int launch_score_planes_queries(const float* Q, int32_t nb, int32_t dim, float* scratch, int mode, hipStream_t s, uint32_t* zero_word = nullptr);
size_t pp_rows_scratch_bytes(int64_t n_rows, int32_t nb, int n_cu, int32_t expected_per_query, int32_t* log_cap_out);
// row tiles (of 128 rows) [tile_begin, tile_begin + tile_count) only (tile_count < 0: to the end); norms_ready: the block norm ranges in
// `work` are those of an earlier launch of the same search
int launch_pp_rows_pass(const void* image, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, const float* row_norm, int mode,
                        const CandArgs* cand, void* work, int32_t log_cap, int n_cu, hipStream_t s, float split_scale, int64_t tile_begin = 0,
                        int64_t tile_count = -1, bool norms_ready = false,
                        bool row_norm_test = false);  // cosine: test hits row by row against their own norms (wild norm spread)
// the sample pass of the same search on that tile (MODE 1): S[q * ld_s + 256 j + r] = similarity of query q with row 256 stride j + r
int launch_pp_rows_sample(const void* image, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, const float* row_norm, int mode, float* S,
                          int64_t ld_s, int32_t stride, int n_cu, hipStream_t s, float split_scale);
// thr[q] = max(thr[q], kth[q * k + k - 1] - window[q]): the k-th best approximate similarity of ANY subset of the rows bounds the k-th best
// overall from below (select.hip; the second round of the fused top-k's candidate pass)
int launch_raise_threshold(float* thr, const float* kth, int32_t nq, int32_t k, const float* window, hipStream_t s);
int launch_score_planes_pass(const void* planes, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, float* scores, int64_t ld,
                             const float* row_norm, const float* row_sumsq, int mode, int32_t tile_stride, const uint32_t* run_if,
                             const CandArgs* cand, int n_cu, hipStream_t s, float split_scale, bool half = false, bool hi_only = false);
// [largest |element|, smallest non-zero row maximum, non-finite flag] of an fp32 corpus, as uint32 bit patterns (device, 3 words)
int launch_row_range(const float* E, int64_t n_rows, int32_t dim, uint32_t* range, hipStream_t s);
// Any dim / nq: one wave per chunk (or per candidate), VALU dot products.
// score_gemm.hip: similarity of many queries at once (fp32 MFMA GEMM, 128 x 128 tiles, fused metric); dim % 32 == 0.
int launch_score_gemm(const float* E, int64_t n_rows, int32_t dim, const float* Q, int32_t nb, float* scores,
                      int64_t ld, const float* row_norm, const float* row_sumsq, float* q_sumsq_scratch, int mode,
                      int n_cu, hipStream_t s, float split_scale = 0.f);
size_t score_gemm_scratch_floats(int32_t nb, int32_t dim, bool split);
// the same over an fp16-stored corpus
int launch_maxsim_stream16(const uint16_t* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                           const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                           float* out, int64_t ld, int n_cu, hipStream_t s);
// exact MaxSim of (query, candidate chunk) pairs, all queries in one launch: dim % 16 == 0 up to 1024, dim % 128 == 0 up to 4096 (wave-private query
// windows, maxsim_pairs_wide_kernel), nq <= 32, fp32 MFMA
int launch_maxsim_pairs(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride, const int64_t* chunk_offsets,
                        const int32_t* candidates, int64_t n_items_per_query, int32_t n_queries, float* out, hipStream_t s,
                        bool rows16 = false,  // rows16: D points at fp16 rows (an fp16-stored corpus)
                        int64_t item_stride = 0, int64_t first_item = 0,  // lists of `item_stride` (0: n_items_per_query) entries per query, of
                                                                          // which entries first_item .. first_item + n_items_per_query - 1 are scored
                        int packed = 1);  // 0: a chunk per tile; 1: rows packed into shared tiles; 2: ... by sixteen waves per workgroup  // dim % 128 == 0: the row-packing kernel (same bits, about half the matrix work on short chunks)
// 1024 < dim <= 4096 (dim % 128 == 0): launch_maxsim_pairs takes maxsim_pairs_wide_kernel; and the exact scores of EVERY chunk behind a run-if flag
int launch_maxsim_pairs_all_wide(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride, const int64_t* offsets, int64_t n_chunks,
                                 int32_t n_queries, float* out, int64_t out_stride, hipStream_t s, const uint32_t* run_if, bool rows16 = false);
int launch_maxsim_generic(const float* D, int32_t dim, const float* Q, int32_t nq, int64_t q_stride_queries,
                          const int64_t* chunk_offsets, const int32_t* candidates, int64_t n_items,
                          int32_t n_queries, float* out, hipStream_t s);
int launch_sanitize_candidates(const int32_t* in, int64_t n, int64_t n_chunks, const uint32_t* live_chunk_bits, int32_t* out,
                               hipStream_t s);
// Rerank fast path: dim == 128, nq <= 32 (MFMA, direct fragment loads).
int launch_maxsim_cand(const float* D, int32_t dim, const float* Q, int32_t nq, const int64_t* chunk_offsets,
                       const int32_t* candidates, int32_t n_cand, int32_t n_queries, float* out, hipStream_t s,
                       float split_scale = 0.f);  // > 0: fp16-split arithmetic, corpus scaled by that power of two
int launch_maxsim_cand16(const uint16_t* D, int32_t dim, const float* Q, int32_t nq, const int64_t* chunk_offsets,
                         const int32_t* candidates, int32_t n_cand, int32_t n_queries, float* out, hipStream_t s);

}  // namespace rl
