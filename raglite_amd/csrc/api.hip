// C ABI of libraglite_hip.so (see include/raglite_hip.h for the contract of every entry point).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "common.h"

namespace rl {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

namespace {

hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Per-thread cache of small device blocks: RL_MEM_HOST calls stage a query in and a handful of results out, and a
// hipMalloc + hipFree pair per staging buffer costs more than a 10 k-row search (measured: 121 us per host-argument
// search_chunks against 39 us with device arguments).  Blocks up to 4 MiB are recycled, larger ones are not kept.
struct BlockCache {
    struct Block { void* p; size_t bytes; int device; };
    std::vector<Block> free_blocks;
    static constexpr size_t MAX_BLOCK = size_t(4) << 20;
    static constexpr size_t MAX_BLOCKS = 32;
    void* take(size_t bytes, int device, size_t* got) {
        for (size_t i = 0; i < free_blocks.size(); ++i) {
            Block b = free_blocks[i];
            if (b.device == device && b.bytes >= bytes && b.bytes <= 4 * bytes + 4096) {
                free_blocks.erase(free_blocks.begin() + (long)i);
                *got = b.bytes;
                return b.p;
            }
        }
        return nullptr;
    }
    void give(void* p, size_t bytes, int device) {
        if (bytes > MAX_BLOCK || free_blocks.size() >= MAX_BLOCKS) { (void)hipFree(p); return; }
        free_blocks.push_back({p, bytes, device});
    }
    ~BlockCache() {
        for (const Block& b : free_blocks) (void)hipFree(b.p);  // at thread exit; errors during runtime teardown are moot
    }
};
thread_local BlockCache g_blocks;

// Per-thread pinned bounce buffer for the small host <-> device copies of RL_MEM_HOST calls (a query in, k results
// out).  A hipMemcpyAsync from / to pageable memory is synchronous and goes through the driver's own staging; through
// pinned memory the copies are truly asynchronous and the call needs ONE synchronisation: results are copied out of
// the pinned buffer after it (drain()).  Entries are keyed by the DevBuf they belong to, so that an early error
// return, which destroys its DevBufs without draining, can never copy stale data into a caller's buffer later.
struct PinnedBounce {
    static constexpr size_t CAP = size_t(1) << 20, SMALL = size_t(256) << 10;
    char* base = nullptr;
    size_t used = 0;
    struct Pending { const void* owner; const void* src; void* dst; size_t bytes; };
    std::vector<Pending> pending;
    void* take(size_t bytes) {
        if (bytes == 0 || bytes > SMALL) return nullptr;
        if (!base && hipHostMalloc(reinterpret_cast<void**>(&base), CAP, hipHostMallocDefault) != hipSuccess) {
            base = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
        const size_t at = (used + 63) & ~size_t(63);
        if (at + bytes > CAP) return nullptr;
        used = at + bytes;
        return base + at;
    }
    void drain() {  // after the stream has been synchronised
        for (const Pending& p : pending) std::memcpy(p.dst, p.src, p.bytes);
        pending.clear();
        used = 0;
    }
    void drop(const void* owner) {
        for (size_t i = pending.size(); i-- > 0;)
            if (pending[i].owner == owner) pending.erase(pending.begin() + (long)i);
    }
    ~PinnedBounce() { if (base) (void)hipHostFree(base); }
};
thread_local PinnedBounce g_pinned;

// Device scratch that returns itself to the thread's block cache; used for RL_MEM_HOST staging and per-call
// temporaries.  Callers synchronise the stream before the buffer goes out of scope whenever the device may still
// be using it (finish() for host calls), so a recycled block is never in flight.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~DevBuf() { release(); }
    void release() {
        g_pinned.drop(this);
        if (p) g_blocks.give(p, bytes, device);
        p = nullptr;
    }
    int alloc(size_t want) {
        release();
        if (want == 0) want = 16;
        RL_HIP(hipGetDevice(&device));
        size_t got = 0;
        p = g_blocks.take(want, device, &got);
        if (p) { bytes = got; return RL_OK; }
        const size_t rounded = (want + 255) & ~size_t(255);
        RL_HIP(hipMalloc(&p, rounded));
        bytes = rounded;
        return RL_OK;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Growable device buffer owned by an index (never shrinks; no allocation in steady state).
struct Pool {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return RL_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        RL_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return RL_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Input staging: returns a device pointer for `src` (copying when it is a host pointer).
template <class T>
int stage_in(const T* src, size_t count, int mem, hipStream_t s, DevBuf& tmp, const T** out) {
    if (mem == RL_MEM_DEVICE) { *out = src; return RL_OK; }
    RL_TRY(tmp.alloc(count * sizeof(T)));
    if (count) {
        const void* from = src;
        if (void* pin = g_pinned.take(count * sizeof(T))) {
            std::memcpy(pin, src, count * sizeof(T));
            from = pin;
        }
        RL_HIP(hipMemcpyAsync(tmp.p, from, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    *out = tmp.as<T>();
    return RL_OK;
}
// Output staging: a device pointer to write to; stage_out copies back for host callers.
template <class T>
int stage_out_begin(T* dst, size_t count, int mem, DevBuf& tmp, T** out) {
    if (mem == RL_MEM_DEVICE || dst == nullptr) { *out = dst; return RL_OK; }
    RL_TRY(tmp.alloc(count * sizeof(T)));
    *out = tmp.as<T>();
    return RL_OK;
}
template <class T>
int stage_out_end(T* dst, size_t count, int mem, hipStream_t s, const DevBuf& tmp) {
    if (mem == RL_MEM_DEVICE || dst == nullptr || count == 0) return RL_OK;
    if (void* pin = g_pinned.take(count * sizeof(T))) {
        RL_HIP(hipMemcpyAsync(pin, tmp.p, count * sizeof(T), hipMemcpyDeviceToHost, s));
        g_pinned.pending.push_back({&tmp, pin, dst, count * sizeof(T)});
        return RL_OK;
    }
    RL_HIP(hipMemcpyAsync(dst, tmp.p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    return RL_OK;
}
// Synchronise and hand the bounced results to the caller's buffers.
int sync_and_drain(hipStream_t s) {
    RL_HIP(hipStreamSynchronize(s));
    g_pinned.drain();
    return RL_OK;
}
int finish(int mem, hipStream_t s) {
    if (mem == RL_MEM_HOST) return sync_and_drain(s);
    return RL_OK;
}

int scan_mode(int metric) {
    switch (metric) {
        case RL_COSINE: return SCAN_COSINE;
        case RL_DOT: return SCAN_DOT;
        case RL_L2: return SCAN_L2;
        default: return -1;
    }
}

constexpr size_t SCORE_BATCH_BYTES = size_t(8) << 30;  // cap of the [B x N] score scratch per sub-batch

// Route options (raglite_hip.h "options"): per index, copied from the process-wide defaults when the index is created.  The library
// reads no environment variable.
struct Options {
    int64_t v[RL_OPT_COUNT_];
    Options() {
        for (auto& x : v) x = 0;
        v[RL_OPT_HI_SEARCH] = v[RL_OPT_HI_MAXSIM] = v[RL_OPT_HI_PRODUCTS] = v[RL_OPT_PP_PASS] = v[RL_OPT_FUSED_TOPK] = v[RL_OPT_FUSED_HI] = 1;
        v[RL_OPT_FUSED_PP] = v[RL_OPT_GEMM_PASS] = v[RL_OPT_QUERY_PAIRS] = v[RL_OPT_PLANES_GEMM] = v[RL_OPT_KEEP_IMAGE] = v[RL_OPT_KEEP_HI] = 1;
        v[RL_OPT_EXACT_KTH_THRESHOLD] = v[RL_OPT_FUSED_TWO_ROUNDS] = v[RL_OPT_KEEP_HI_PLANE] = v[RL_OPT_F16_EXACT] = v[RL_OPT_LAZY_IMAGES] = v[RL_OPT_FUSED_PP_SAMPLE] = v[RL_OPT_LIST_SELECT] = v[RL_OPT_HI_FEW] = 1;
        v[RL_OPT_TOPK_BLOCK] = v[RL_OPT_PAIRS_PACKED] = 2;
        v[RL_OPT_HI_PIVOT] = 1;
        v[RL_OPT_IMAGE_HEADROOM_MB] = -1;
        v[RL_OPT_ARITHMETIC] = RL_ARITH_AUTO;
    }
    bool on(int key) const { return v[key] != 0; }
};
std::mutex g_default_opts_mu;
Options g_default_opts;
bool option_value_ok(int key, int64_t value) {
    switch (key) {
        case RL_OPT_HI_SEARCH: case RL_OPT_HI_MAXSIM: case RL_OPT_PP_PASS: case RL_OPT_FUSED_TOPK: case RL_OPT_FUSED_HI: case RL_OPT_FUSED_PP:
        case RL_OPT_GEMM_PASS: case RL_OPT_QUERY_PAIRS: case RL_OPT_PLANES_GEMM: case RL_OPT_KEEP_IMAGE: case RL_OPT_KEEP_HI:
        case RL_OPT_EXACT_KTH_THRESHOLD: case RL_OPT_FUSED_TWO_ROUNDS: case RL_OPT_KEEP_HI_PLANE: case RL_OPT_F16_EXACT: case RL_OPT_LAZY_IMAGES: case RL_OPT_FUSED_PP_SAMPLE: case RL_OPT_LIST_SELECT: case RL_OPT_HI_FEW: case RL_OPT_HI_PIVOT:
            return value == 0 || value == 1;
        case RL_OPT_TOPK_BLOCK: case RL_OPT_PAIRS_PACKED: return value >= 0 && value <= 2;
        case RL_OPT_HI_PRODUCTS: return value == 1 || value == 2;
        case RL_OPT_FUSED_TOPK_CAP: return value >= 0 && value <= MERGE_CAP;
        case RL_OPT_FUSED_TOPK_STRIDE: return value == 0 || (value >= 2 && value <= (int64_t(1) << 20));
        case RL_OPT_IMAGE_HEADROOM_MB: return value >= -1 && value <= (int64_t(1) << 30);
        case RL_OPT_ARITHMETIC: return value == RL_ARITH_AUTO || value == RL_ARITH_FP32_EXACT;
        default: return false;
    }
}

}  // namespace
}  // namespace rl

struct rl_index {
    const float* E = nullptr;       // fp32 storage ...
    const uint16_t* E16 = nullptr;  // ... or IEEE fp16 storage (rl_index_create_f16); exactly one is set
    bool owns_E = false;
    int64_t n_rows = 0;
    int32_t dim = 0;
    int64_t n_chunks = 0;
    int metric = RL_COSINE;
    bool has_empty_chunk = false;
    int64_t* offsets = nullptr;       // device [n_chunks + 1]
    int32_t* row_to_chunk = nullptr;  // device [n_rows + 65]
    float* norm = nullptr;            // device [n_rows]  (cosine)
    float* sumsq = nullptr;           // device [n_rows]  (l2)
    int n_cu = 256;
    std::mutex mu;
    rl::SelectWorkspace ws;
    rl::Pool scores;                  // [B x ld] similarity scratch / chunk scores
    rl::Pool hits;                    // search_chunks: [B x num_hits] (score, row)
    rl::Pool misc;
    // lifecycle (append / delete / filter)
    int64_t cap_rows = 0;                 // rows the owned buffers (E, norm, sumsq, row_to_chunk) can hold
    int64_t cap_chunks = 0;               // chunks the device CSR can hold
    std::vector<int64_t> h_offsets;       // host copy of the CSR
    std::vector<uint32_t> h_live;         // host bitset over chunks: 1 = live (empty until the first delete)
    uint32_t* live_chunk_bits = nullptr;  // device copy of h_live (nullptr: every chunk is live)
    uint32_t* live_row_bits = nullptr;    // the same expanded to rows
    int64_t n_dead_chunks = 0, n_dead_rows = 0;
    rl::Pool maskbuf;                     // per-call effective row mask
    rl::Pool qsplit;                      // fp16 (hi, lo) query fragments of a MaxSim batch (maxsim_stream.hip)
    // SPLIT arithmetic of the stream kernel (fp32 storage only): range of the row norms, and what follows from it
    uint32_t* d_range = nullptr;          // device scratch of launch_row_range
    float max_abs = 0.f, min_row_max = std::numeric_limits<float>::infinity();  // over rows that are not all zero
    bool nonfinite = false;
    int arithmetic = RL_ARITH_AUTO;
    float split_scale = 0.f;              // > 0: power of two applied to the corpus inside the kernel; 0: exact fp32 MFMAs
    // Pre-split corpus image of maxsim_gemm.hip (fp16 hi | lo planes in the kernel's LDS layout, 4 B per element) and
    // the "last row of its chunk" bitmap; built with the index, extended on append, rebuilt when split_scale changes.
    rl::Pool planes, ends, qplanes;
    rl::Pool q32;                         // rl_maxsim_topk_batch_f16: the fp16 queries widened to fp32 (what the query-side kernels read)
    rl::Pool cand;                        // rl_maxsim_rerank: sanitised candidate ordinals
    rl::Pool fused;                       // fused batched top-k: sample scores, thresholds, candidate lists, counters
    rl::Pool pp_work;                     // ... on the sixteen-group tile: wave-private record logs, block norm ranges (maxsim_pp.hip MODE 2)
    rl::Pool rankbuf;                     // rank cut (order-first-then-filter): histogram levels + tie counts
    // HI plane (round 2): fp16(e * split_scale) rounded to nearest (toward zero until round 3), row-major [n_rows x dim] -- the hi halves of the fp16
    // split as a matrix of their own, 2 B per element: what the single-query search streams (search_rows_hi).
    rl::Pool hiplane, hibuf;
    float hi_scale = 0.f;                 // the scale the plane was built with; 0 = no plane
    int64_t hi_rows = 0;                  // rows it covers
    // ... and the same halves in the one-plane IMAGE layout (maxsim_gemm.hip HALF): what the approximate MaxSim pass of a
    // batch multiplies (maxsim_batch_hi); max_row_norm = max |e| over the rows, for its error bound
    rl::Pool hi_image;
    float hi_image_scale = 0.f;
    int64_t hi_image_rows = 0;
    float max_row_norm = 0.f, max_lo_norm = 0.f, max_lo_ratio = 0.f;  // max |e|, max |e_lo|, max |e_lo| / |e| (e_lo: what the HI halves drop)
    float min_row_norm = std::numeric_limits<float>::infinity();  // min |e| over the rows folded in so far (never raised by deletions: conservative)
    float max_row_norm_scale = 0.f;       // the split scale they were computed at
    uint32_t* d_norms = nullptr;          // device scratch of launch_max_row_norm (4 words)
    int64_t max_row_norm_rows = 0;        // rows folded into them
    // The scratch above is shared by all calls on this handle; `mu` serialises only their host side.  Device-mode calls are
    // asynchronous, so a call arriving on a DIFFERENT stream than the previous one first waits for that stream.
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
    int64_t ends_rows = -1;               // rows the `ends` bitmap covers (-1: not built); kept by both images
    float planes_scale = 0.f;             // the scale the image was built with; 0 = no image
    int64_t planes_rows = 0;              // rows the image covers
    // What the last bound-filtered search on this handle left behind (rl_index_filter_stats): per-query candidate counters and the
    // device flag its guarded full-precision fallback waits on.  Pointers into the scratch above, valid until the next call.
    // rl_rank_cut_*: the staged rank cut of a SHARDED corpus keeps its queries and scores here between the calls
    int32_t rank_B = 0;                   // queries of the running rl_rank_cut_begin (0: none)
    int32_t mb_B = 0, mb_nq = 0, mb_k = 0;  // rl_maxsim_batch_begin in progress: queries, vectors per query, k (0: none)
    uint64_t scratch_epoch = 0, mb_epoch = 0;  // calls that used the scratch so far; the value right after that rl_maxsim_batch_begin
    rl::Pool rank_q;                      // their device copy (the l2 re-scoring of rl_rank_cut_finish needs them)
    struct FilterRecord { int kind = 0; int32_t n = 0, cap = 0; const uint32_t* cnt = nullptr; const uint32_t* flag = nullptr; } filt;
    rl::Options opt;                      // route options (rl_index_set_option)
    // RL_OPT_LAZY_IMAGES (round 5, the default): an image is built by the first call whose route reads it -- which images this index
    // has been asked for so far (IMG_* bits); with the option off every image the KEEP_* options allow is built with the index
    uint32_t demanded = 0;
    uint32_t no_room = 0;                 // IMG_* bits whose last build was skipped because it would not have left the headroom free
    uint64_t no_room_retry_epoch = 0;     // ... and the scratch epoch from which demand_images asks for them again
    // ... and a pinned host word every bound-filtered MaxSim batch copies its fallback flag to when it is done (asynchronously): a batch
    // that finds the previous one fell back asks for the pre-split image, so that an index whose data keeps defeating the bound runs its
    // full-precision passes through the eight-query kernel instead of the streaming kernels (3-4 x faster) from the second batch on
    uint32_t* h_fell_back = nullptr;
    uint32_t* d_fell_back = nullptr;  // the device's view of that word (written by guarded_select_kernel)
    // rl_time_kernel kind 8: what the candidate pass of the last fused-HI row search ran with (pointers into misc / fused / pp_work: valid
    // while those pools have not been re-reserved, which `pools` pins down)
    struct FusedReplay {
        bool valid = false, pp = false, row_test = false; int32_t B = 0, log_cap = 0; int mode = 0; float* qs = nullptr; rl::CandArgs ca{}; uint32_t* cnt = nullptr;
        const float* thr1 = nullptr; int64_t round1_tiles = 0;  // two-round candidate pass: the first round's thresholds and tiles
        const void* pools[3] = {nullptr, nullptr, nullptr};
    } replay;
};

namespace {
// Called (under idx->mu) by every entry point that uses the index' shared scratch: a call on another stream than the
// previous one waits for the previous stream's work on this handle (host-side; the rare case).
int use_scratch(rl_index* idx, hipStream_t s) {
    ++idx->scratch_epoch;  // (staged calls check that nothing else used the scratch between their stages)
    idx->filt = {};        // its pointers go into scratch this call may re-reserve: whoever filters next records itself again
    idx->ws.block_route = (int)idx->opt.v[RL_OPT_TOPK_BLOCK];
    idx->replay.valid = false;  // likewise: a later search overwrites the queries / thresholds the record points at (rl_time_kernel restores it for itself)
    if (idx->last_stream_set && idx->last_stream != s) RL_HIP(hipStreamSynchronize(idx->last_stream));
    idx->last_stream = s;
    idx->last_stream_set = true;
    return RL_OK;
}

// The stream kernel multiplies an fp32 corpus either with the exact fp32 MFMA chain or -- 10 % faster, HBM-bound instead
// of matrix-pipe-bound -- as fp16 (hi, lo) pairs (maxsim_stream.hip, SPLIT).  After scaling the largest element to
// [2^13, 2^14) a pair keeps 22 significant bits of every element down to 2^-3, fewer below (lo goes subnormal), which
// is harmless inside a row (the loss is < 2^-38 of the row's largest element) but not for a whole row far below the
// rest.  So SPLIT is used only when the largest elements of all non-zero rows lie within a factor 2^10 of each other
// (every normalised corpus does), everything is finite, and the caller has not asked for the exact chain.
void update_split_scale(rl_index* idx) {
    idx->split_scale = 0.f;
    if (idx->E16 || !idx->E || idx->arithmetic == RL_ARITH_FP32_EXACT || idx->nonfinite) return;
    if (idx->dim % 4) return;
    if (idx->max_abs == 0.f) { idx->split_scale = 1.f; return; }  // all-zero (or empty) corpus: nothing to lose
    if (!(idx->max_abs / idx->min_row_max <= 1024.f)) return;      // row magnitudes spread over more than 2^10
    int ex = 0;
    (void)std::frexp(idx->max_abs, &ex);                           // largest |e| = f * 2^ex, f in [0.5, 1)
    idx->split_scale = std::ldexp(1.f, 14 - std::max(-100, ex));   // |e| * scale < 2^14
}

// Folds the magnitude range of rows [first, first + n) into the index (synchronises the stream: build / append only).
int scan_row_range(rl_index* idx, int64_t first, int64_t n, hipStream_t s) {
    if (idx->E16 || !idx->E || idx->dim % 4) return RL_OK;
    if (!idx->d_range) RL_HIP(hipMalloc(&idx->d_range, 16));
    RL_TRY(rl::launch_row_range(idx->E + (size_t)first * idx->dim, n, idx->dim, idx->d_range, s));
    uint32_t h[3];
    RL_HIP(hipMemcpyAsync(h, idx->d_range, sizeof(h), hipMemcpyDeviceToHost, s));
    RL_HIP(hipStreamSynchronize(s));
    float mx, mn;
    std::memcpy(&mx, &h[0], 4);
    std::memcpy(&mn, &h[1], 4);
    idx->max_abs = std::max(idx->max_abs, mx);
    idx->min_row_max = std::min(idx->min_row_max, mn);
    idx->nonfinite |= h[2] != 0;
    update_split_scale(idx);
    return RL_OK;
}

// Scale of the corpus image the index should have: the split scale of an fp32 corpus, 1 for an fp16-stored one (its image is
// the stored halves, permuted), 0 = no image.
float image_scale(const rl_index* idx) { return idx->E16 ? 1.0f : idx->split_scale; }
// Which of the three images may exist right now: all of them, or -- lazy images -- those some call has asked for (demand_images)
enum : uint32_t { IMG_PLANES = 1u, IMG_HI_IMAGE = 2u, IMG_HI_PLANE = 4u, IMG_ALL = 7u };
uint32_t image_need(const rl_index* idx) { return idx->opt.on(RL_OPT_LAZY_IMAGES) ? idx->demanded : IMG_ALL; }
bool image_valid(const rl_index* idx) {
    return idx->planes_scale > 0.f && idx->planes_scale == image_scale(idx) && idx->planes_rows == idx->n_rows && idx->n_rows > 0;
}

// The images are optional accelerators (pre-split image 4 B, HI image 2 B, HI plane 2 B per element next to the rows): one is only
// built while it leaves `image_headroom()` of the device free for the per-call scratch (score batches up to 8 GB, selection workspace,
// candidate lists) and for the caller -- an index close to the device's capacity searches through the kernels over the stored rows
// instead of failing an allocation in the middle of a search.  RL_OPT_IMAGE_HEADROOM_MB overrides the default
// max(2 GiB, 1/16 of the device); rl_index_memory reports what was built.
size_t image_headroom(const rl_index* idx) {
    const int64_t mb = idx->opt.v[RL_OPT_IMAGE_HEADROOM_MB];
    if (mb >= 0) return (size_t)mb << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return (size_t)2 << 30; }
    return std::max<size_t>((size_t)2 << 30, total_b / 16);
}
bool image_fits(const rl_index* idx, const rl::Pool& pool, size_t need) {
    if (pool.cap >= need) return true;  // already paid for: nothing has to grow
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return true; }
    return free_b + pool.cap >= need + image_headroom(idx);  // (reserve() frees the old block before it allocates)
}

// Builds / extends the corpus image so that it covers rows [0, idx->n_rows) at image_scale(idx).  Not having the image is
// never an error (the streaming kernels read the stored rows): an allocation failure or a device too full just leaves it absent.

int refresh_row_norm16(rl_index* idx, hipStream_t s);
// "last row of its chunk" bitmap over the current rows: what the batch kernels over either image find chunk ends with (125 KB per 1 M rows)
int refresh_ends(rl_index* idx, hipStream_t s) {
    if (idx->ends_rows == idx->n_rows && idx->ends.p) return RL_OK;
    const int64_t cap = std::max<int64_t>(idx->n_rows, idx->owns_E ? idx->cap_rows : idx->n_rows);
    RL_TRY(idx->ends.reserve(rl::chunk_ends_words(cap) * sizeof(uint32_t)));
    RL_TRY(rl::launch_chunk_ends(idx->row_to_chunk, idx->n_rows, idx->ends.as<uint32_t>(), s));
    idx->ends_rows = idx->n_rows;
    return RL_OK;
}
int refresh_planes(rl_index* idx, hipStream_t s) {
    const bool half = idx->E16 != nullptr;
    const bool want = idx->opt.on(RL_OPT_KEEP_IMAGE) && (idx->E16 || idx->E) && image_scale(idx) > 0.f && idx->dim % 32 == 0 && idx->dim >= 32 &&
                      idx->n_rows > 0 && (image_need(idx) & IMG_PLANES);
    idx->ends_rows = -1;  // (rows or chunk structure may have changed: whoever needs the bitmap rebuilds it)
    if (!want) {
        idx->planes.release();
        idx->planes_scale = 0.f;
        idx->planes_rows = 0;
        return RL_OK;
    }
    const int64_t cap = std::max<int64_t>(idx->n_rows, idx->owns_E ? idx->cap_rows : idx->n_rows);
    const size_t need = rl::planes_bytes(cap, idx->dim, half);
    int64_t first = idx->planes_scale == image_scale(idx) ? (idx->planes_rows & ~int64_t(15)) : 0;
    if (idx->planes.cap < need) first = 0;  // Pool::reserve does not keep the contents
    // (an exactly sized image is "already paid for" and must survive an append into spare capacity however full the device has become)
    if (!image_fits(idx, idx->planes, need) || idx->planes.reserve(need) != RL_OK) {
        (void)hipGetLastError();
        idx->no_room |= IMG_PLANES;
        idx->planes.release();
        idx->planes_scale = 0.f;
        idx->planes_rows = 0;
        return RL_OK;
    }
    const int st = half ? rl::launch_preformat_rows16(idx->E16, first, idx->n_rows, idx->dim, idx->planes.p, s)
                        : rl::launch_presplit_rows(idx->E, first, idx->n_rows, idx->dim, idx->split_scale, idx->planes.p, s);
    if (st == RL_ERR_UNSUPPORTED) {  // e.g. caller-owned rows that are not 16-byte aligned: no image, the streaming kernels serve
        idx->planes.release();
        idx->planes_scale = 0.f;
        idx->planes_rows = 0;
        return RL_OK;
    }
    RL_TRY(st);
    if (refresh_ends(idx, s) != RL_OK) {  // (the bitmap could not be allocated: no image either -- "never an error", as above)
        (void)hipGetLastError();
        idx->planes.release();
        idx->planes_scale = 0.f;
        idx->planes_rows = 0;
        return RL_OK;
    }
    idx->planes_scale = image_scale(idx);
    idx->planes_rows = idx->n_rows;
    if (half) RL_TRY(refresh_row_norm16(idx, s));
    return RL_OK;
}

// The HI plane follows the corpus like the image does: built for big fp32 corpora in split arithmetic whose dim the fp16
// stream kernel takes; RAGLITE_NO_HI_PLANE=1 disables it (2 B per element of extra HBM).
// The dims the half-bytes routes take: any multiple of 32 up to 1024; beyond -- the 1536- to 4096-wide embedders the reference also accepts
// (src/raglite/_embed.py:155-158) -- multiples of 128 (round 6: the exact re-scoring kernels walk wider queries in 128-column windows).
constexpr int32_t HI_MAX_DIM = 4096;
bool hi_dim_ok(int32_t d) { return d % 32 == 0 && d >= 32 && (d <= 1024 || (d <= HI_MAX_DIM && d % 128 == 0)); }
// The rounding term of every half-bytes bound: an fp32 sum of `dim` products is off by at most dim 2^-24 of sum |products| <= |q| |e|, once in
// the approximate pass and once in the exact one, + the query's 2^-22 split: 2^-12 |q| |e| per started 1024 terms (twice what it takes).
float sum_eps(int32_t d) { return 0x1p-12f * (float)((d + 1023) / 1024); }
bool hi_valid(const rl_index* idx) {
    return idx->hi_scale > 0.f && idx->hi_scale == idx->split_scale && idx->hi_rows == idx->n_rows && idx->n_rows > 0;
}
bool hi_image_valid(const rl_index* idx) {
    return idx->hi_image_scale > 0.f && idx->hi_image_scale == idx->split_scale && idx->hi_image_rows == idx->n_rows && idx->n_rows > 0 &&
           idx->max_row_norm_rows == idx->n_rows && idx->max_row_norm > 0.f;
}
// What the approximate MaxSim pass of a batch multiplies at ONE fp16 product per multiply: the image of the hi halves of an fp32 corpus --
// or the image of an fp16-STORED corpus itself (the stored halves ARE the corpus: the pass then drops only the queries' lo halves, the
// bound has no e_lo term, and the candidates are re-scored over the stored rows).  Same size gate for both (>= 64 M elements).
bool approx_image_valid(const rl_index* idx) {
    if (!idx->E16) return hi_image_valid(idx);
    return image_valid(idx) && (int64_t)idx->n_rows * idx->dim >= (int64_t(64) << 20) && hi_dim_ok(idx->dim) && idx->max_row_norm_rows == idx->n_rows &&
           idx->max_row_norm > 0.f;
}
const void* approx_image(const rl_index* idx) { return idx->E16 ? idx->planes.p : idx->hi_image.p; }
float approx_scale(const rl_index* idx) { return idx->E16 ? 1.0f : idx->split_scale; }

// max |e| over the rows of an fp16-stored corpus, folded in as rows arrive (synchronises the stream: build / append / compact only)
int refresh_row_norm16(rl_index* idx, hipStream_t s) {
    const bool off = !idx->opt.on(RL_OPT_KEEP_HI);  // (the switch of the half-bytes paths)
    if (!idx->E16 || off || !image_valid(idx) || (int64_t)idx->n_rows * idx->dim < (int64_t(64) << 20)) {
        if (idx->E16) { idx->max_row_norm = 0.f; idx->max_row_norm_rows = 0; }
        return RL_OK;
    }
    if (idx->max_row_norm_rows > idx->n_rows) { idx->max_row_norm = 0.f; idx->max_row_norm_rows = 0; }  // (compacted: start over)
    if (idx->max_row_norm_rows == idx->n_rows) return RL_OK;
    const int64_t from = idx->max_row_norm_rows;
    if (!idx->d_norms) RL_HIP(hipMalloc(&idx->d_norms, 16));
    uint32_t bits[4] = {0, 0, 0, 0};
    std::memcpy(&bits[0], &idx->max_row_norm, 4);
    RL_HIP(hipMemcpyAsync(idx->d_norms, bits, 16, hipMemcpyHostToDevice, s));
    RL_TRY(rl::launch_max_row_norm16(idx->E16 + (size_t)from * idx->dim, idx->n_rows - from, idx->dim, idx->d_norms, s));
    RL_HIP(hipMemcpyAsync(bits, idx->d_norms, 16, hipMemcpyDeviceToHost, s));
    RL_HIP(hipStreamSynchronize(s));
    std::memcpy(&idx->max_row_norm, &bits[0], 4);
    idx->max_lo_norm = idx->max_lo_ratio = 0.f;
    idx->max_row_norm_rows = idx->n_rows;
    idx->max_row_norm_scale = 1.0f;
    return RL_OK;
}

// max |e|, max |e_lo|, max |e_lo| / |e|, min |e| over the rows at the current split scale: what the error bounds of BOTH half-bytes routes are
// made of (the HI image's and the HI plane's), folded in as rows arrive (synchronises the stream: image builds only)
int refresh_hi_norms(rl_index* idx, hipStream_t s) {
    if (idx->max_row_norm_rows != idx->n_rows || idx->max_row_norm_scale != idx->split_scale) {  // fold the new rows' norms in
        if (idx->max_row_norm_scale != idx->split_scale) {  // (the dropped halves depend on the scale: start over)
            idx->max_row_norm = idx->max_lo_norm = idx->max_lo_ratio = 0.f;
            idx->min_row_norm = std::numeric_limits<float>::infinity();
            idx->max_row_norm_rows = 0;
        }
        const int64_t from = std::min<int64_t>(idx->max_row_norm_rows, idx->n_rows);
        if (!idx->d_norms) RL_HIP(hipMalloc(&idx->d_norms, 16));
        uint32_t bits[4] = {0, 0, 0, 0};
        std::memcpy(&bits[0], &idx->max_row_norm, 4);
        std::memcpy(&bits[1], &idx->max_lo_norm, 4);
        std::memcpy(&bits[2], &idx->max_lo_ratio, 4);
        std::memcpy(&bits[3], &idx->min_row_norm, 4);
        RL_HIP(hipMemcpyAsync(idx->d_norms, bits, 16, hipMemcpyHostToDevice, s));
        RL_TRY(rl::launch_max_row_norm(idx->E + (size_t)from * idx->dim, idx->n_rows - from, idx->dim, idx->split_scale, idx->d_norms, s));
        RL_HIP(hipMemcpyAsync(bits, idx->d_norms, 16, hipMemcpyDeviceToHost, s));
        RL_HIP(hipStreamSynchronize(s));
        std::memcpy(&idx->max_row_norm, &bits[0], 4);
        std::memcpy(&idx->max_lo_norm, &bits[1], 4);
        std::memcpy(&idx->max_lo_ratio, &bits[2], 4);
        std::memcpy(&idx->min_row_norm, &bits[3], 4);
        idx->max_row_norm_rows = idx->n_rows;
        idx->max_row_norm_scale = idx->split_scale;
    }
    return RL_OK;
}
// The HI halves in image layout + the largest row norm (synchronises the stream: build / append / compact only).
int refresh_hi_image(rl_index* idx, hipStream_t s) {
    const bool off = !idx->opt.on(RL_OPT_KEEP_HI);  // (shared with the row-major plane)
    // (round 4: independent of the pre-split image -- an index with RL_OPT_KEEP_IMAGE = 0 keeps rows + HI image, 1.5 x the corpus, and its
    // MaxSim batches fall back to the streaming kernels over the rows)
    const bool want = !off && !idx->E16 && idx->E && idx->split_scale > 0.f && hi_dim_ok(idx->dim) &&
                      (int64_t)idx->n_rows * idx->dim >= (int64_t(64) << 20) && (image_need(idx) & IMG_HI_IMAGE);
    if (!want) {
        idx->hi_image.release();
        idx->hi_image_scale = 0.f;
        idx->hi_image_rows = 0;
        return RL_OK;
    }
    const int64_t cap = std::max<int64_t>(idx->n_rows, idx->owns_E ? idx->cap_rows : idx->n_rows);
    const size_t need = rl::planes_bytes(cap, idx->dim, true);
    int64_t first = idx->hi_image_scale == idx->split_scale ? (idx->hi_image_rows & ~int64_t(15)) : 0;
    if (idx->hi_image.cap < need) first = 0;
    if (!image_fits(idx, idx->hi_image, need) || idx->hi_image.reserve(need) != RL_OK) {
        (void)hipGetLastError();
        idx->no_room |= IMG_HI_IMAGE;
        idx->hi_image.release();
        idx->hi_image_scale = 0.f;
        idx->hi_image_rows = 0;
        return RL_OK;
    }
    const int st = rl::launch_presplit_hi_rows(idx->E, first, idx->n_rows, idx->dim, idx->split_scale, idx->hi_image.p, s);
    if (st == RL_ERR_UNSUPPORTED) {
        idx->hi_image.release();
        idx->hi_image_scale = 0.f;
        idx->hi_image_rows = 0;
        return RL_OK;
    }
    RL_TRY(st);
    if (refresh_ends(idx, s) != RL_OK) {  // (as for the pre-split image: without the bitmap the image is just absent)
        (void)hipGetLastError();
        idx->hi_image.release();
        idx->hi_image_scale = 0.f;
        idx->hi_image_rows = 0;
        return RL_OK;
    }
    idx->hi_image_scale = idx->split_scale;
    idx->hi_image_rows = idx->n_rows;
    RL_TRY(refresh_hi_norms(idx, s));
    return RL_OK;
}
int refresh_hi_plane(rl_index* idx, hipStream_t s) {
    RL_TRY(refresh_hi_image(idx, s));
    const bool off = !idx->opt.on(RL_OPT_KEEP_HI) || !idx->opt.on(RL_OPT_KEEP_HI_PLANE);
    const int32_t d = idx->dim;
    // (the stream kernel's dims; wider embedders -- round 6 -- go through the packed scan, scan16.hip)
    const bool dim_ok = d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024 || (d > 1024 && hi_dim_ok(d));
    const bool want = !off && !idx->E16 && idx->E && idx->split_scale > 0.f && dim_ok &&
                      (int64_t)idx->n_rows * d >= (int64_t(64) << 20) && (idx->metric == RL_COSINE || idx->metric == RL_DOT || idx->metric == RL_L2) &&
                      (image_need(idx) & IMG_HI_PLANE);
    if (!want) {
        idx->hiplane.release();
        idx->hi_scale = 0.f;
        idx->hi_rows = 0;
        return RL_OK;
    }
    const int64_t cap = std::max<int64_t>(idx->n_rows, idx->owns_E ? idx->cap_rows : idx->n_rows);
    const size_t need = (size_t)cap * d * sizeof(uint16_t);
    int64_t first = idx->hi_scale == idx->split_scale ? idx->hi_rows : 0;
    if (idx->hiplane.cap < need) first = 0;  // Pool::reserve does not keep the contents
    if (!image_fits(idx, idx->hiplane, need) || idx->hiplane.reserve(need) != RL_OK) {
        (void)hipGetLastError();
        idx->no_room |= IMG_HI_PLANE;
        idx->hiplane.release();
        idx->hi_scale = 0.f;
        idx->hi_rows = 0;
        return RL_OK;
    }
    const int st = rl::launch_cast_f16_scaled(idx->E + (size_t)first * d, idx->hiplane.as<uint16_t>() + (size_t)first * d,
                                           (idx->n_rows - first) * d, idx->split_scale, s);
    if (st == RL_ERR_UNSUPPORTED) {  // caller-owned rows that are not 16-byte aligned
        idx->hiplane.release();
        idx->hi_scale = 0.f;
        idx->hi_rows = 0;
        return RL_OK;
    }
    RL_TRY(st);
    RL_TRY(refresh_hi_norms(idx, s));  // (the plane's bound needs them whether or not the HI image exists)
    idx->hi_scale = idx->split_scale;
    idx->hi_rows = idx->n_rows;
    return RL_OK;
}

// Lazy images: called (under idx->mu, on the call's stream) by every route that reads an image, BEFORE it tests the image's validity.  The
// first call that asks for an image builds it (allocation + one pass over the rows: ~2 ms per image at 1 M x 1024, synchronous); later
// calls find the bit set.  What an index has built is a function of the calls it has served: a MaxSim-only deployment keeps rows + HI
// image (1.5 x the corpus), a single-query search service rows + HI plane (1.5 x), batches of >= 96 row queries add the pre-split image.
int demand_images(rl_index* idx, uint32_t bits, hipStream_t s) {
    if (!idx->opt.on(RL_OPT_LAZY_IMAGES) || (idx->demanded & bits) == bits) return RL_OK;
    const uint32_t fresh = bits & ~idx->demanded;
    if ((fresh & idx->no_room) && idx->scratch_epoch < idx->no_room_retry_epoch && !(fresh & ~idx->no_room)) return RL_OK;  // (asked recently, no room then)
    idx->demanded |= bits;
    idx->no_room &= ~fresh;
    if (fresh & IMG_PLANES) RL_TRY(refresh_planes(idx, s));
    if (fresh & (IMG_HI_IMAGE | IMG_HI_PLANE)) RL_TRY(refresh_hi_plane(idx, s));
    // An image that was NOT built because the device was too full at this moment (scratch pools and the caller's allocator have grown
    // since the index was created) must not leave the route on its slow path for good: its demand bit is cleared again, and a later
    // call -- at most one in 64, a hipMemGetInfo each -- asks again.  (An image the options / the shape do not allow stays "demanded".)
    if (const uint32_t retry = fresh & idx->no_room) {
        idx->demanded &= ~retry;
        idx->no_room_retry_epoch = idx->scratch_epoch + 64;
    }
    return RL_OK;
}
// the image the approximate MaxSim pass of a batch multiplies: the HI image of an fp32 index, the (only) image of an fp16-stored one
uint32_t approx_image_bit(const rl_index* idx) { return idx->E16 ? IMG_PLANES : IMG_HI_IMAGE; }
}  // namespace

using namespace rl;

extern "C" {

int rl_version(void) { return 100; }
const char* rl_last_error(void) { return g_last_error.c_str(); }

int rl_init(int device) {
    int n = 0;
    RL_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RL_ERR_INVALID, "rl_init: no such device");
    RL_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RL_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RL_ERR_UNSUPPORTED, std::string("rl_init: libraglite_hip is built for gfx950 only, found ") +
                                            prop.gcnArchName);
    return RL_OK;
}

int rl_device_count(int* count) {
    if (!count) return fail(RL_ERR_INVALID, "rl_device_count: null argument");
    RL_HIP(hipGetDeviceCount(count));
    return RL_OK;
}

int rl_device_info(int device, char* name, int len, int* compute_units, int64_t* total_mem) {
    hipDeviceProp_t prop;
    RL_HIP(hipGetDeviceProperties(&prop, device));
    if (name && len > 0) {
        std::strncpy(name, prop.gcnArchName, (size_t)len - 1);
        name[len - 1] = 0;
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
    return RL_OK;
}

int rl_dev_alloc(void** ptr, size_t bytes) {
    if (!ptr) return fail(RL_ERR_INVALID, "rl_dev_alloc: null argument");
    RL_HIP(hipMalloc(ptr, bytes ? bytes : 16));
    return RL_OK;
}
int rl_dev_free(void* ptr) {
    if (ptr) RL_HIP(hipFree(ptr));
    return RL_OK;
}
int rl_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes) RL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return RL_OK;
}
int rl_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes) RL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    RL_HIP(hipStreamSynchronize(as_stream(stream)));
    return RL_OK;
}
int rl_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes) RL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return RL_OK;
}
int rl_stream_sync(void* stream) {
    RL_HIP(hipStreamSynchronize(as_stream(stream)));
    return RL_OK;
}

int rl_synth_fill(float* dst, int64_t start, int64_t count, uint64_t seed, int kind, void* stream) {
    if (count < 0 || (count > 0 && !dst)) return fail(RL_ERR_INVALID, "rl_synth_fill: bad arguments");
    if (kind != RL_SYNTH_UNIFORM && kind != RL_SYNTH_SMALL_INT) return fail(RL_ERR_INVALID, "rl_synth_fill: bad kind");
    return launch_synth(dst, start, count, seed, kind, as_stream(stream));
}

// ---- a1 + a2 + a3 ----------------------------------------------------------------------------------
int rl_pool_norm(const float* tokens, int64_t n_token_rows, int32_t dim, const int64_t* span_begin,
                 const int64_t* span_end, int64_t n_spans, int32_t normalize, double eps, float* out_f32,
                 uint16_t* out_f16, int mem, void* stream) {
    if (dim <= 0 || n_token_rows < 0 || n_spans < 0) return fail(RL_ERR_INVALID, "rl_pool_norm: negative size");
    if (n_spans > 0 && (!span_begin || !span_end)) return fail(RL_ERR_INVALID, "rl_pool_norm: null spans");
    if (!out_f32 && !out_f16) return fail(RL_ERR_INVALID, "rl_pool_norm: no output requested");
    if (!(eps >= 0.0)) return fail(RL_ERR_INVALID, "rl_pool_norm: eps must be >= 0");
    if (n_spans == 0) return RL_OK;
    hipStream_t s = as_stream(stream);
    if (mem == RL_MEM_HOST) {  // validate the spans the host handed over (device callers own their spans)
        for (int64_t i = 0; i < n_spans; ++i)
            if (span_begin[i] < 0 || span_end[i] < span_begin[i] || span_end[i] > n_token_rows)
                return fail(RL_ERR_INVALID, "rl_pool_norm: span outside the token matrix");
    }
    DevBuf t_tok, t_b, t_e, t_o32, t_o16;
    const float* d_tok; const int64_t* d_b; const int64_t* d_e;
    RL_TRY(stage_in(tokens, (size_t)n_token_rows * dim, mem, s, t_tok, &d_tok));
    RL_TRY(stage_in(span_begin, (size_t)n_spans, mem, s, t_b, &d_b));
    RL_TRY(stage_in(span_end, (size_t)n_spans, mem, s, t_e, &d_e));
    float* d_o32; uint16_t* d_o16;
    RL_TRY(stage_out_begin(out_f32, (size_t)n_spans * dim, mem, t_o32, &d_o32));
    RL_TRY(stage_out_begin(out_f16, (size_t)n_spans * dim, mem, t_o16, &d_o16));
    RL_TRY(launch_pool_norm(d_tok, dim, d_b, d_e, n_spans, normalize, eps, d_o32, d_o16, s));
    RL_TRY(stage_out_end(out_f32, (size_t)n_spans * dim, mem, s, t_o32));
    RL_TRY(stage_out_end(out_f16, (size_t)n_spans * dim, mem, s, t_o16));
    return finish(mem, s);
}

// Below this many queries the stream kernel's per-32-query corpus passes (HBM-bound) beat a 128-query GEMM tile.
constexpr int32_t GEMM_MIN_QUERIES = 96;
// (a WIDE index -- dim > 1024 -- has no 32-queries-per-pass streaming kernel between the few-queries routes and the GEMM-shaped ones: its scan takes ONE
// query per pass, so the GEMM-shaped routes start right behind the few-queries search there)
int32_t rows_gemm_min(const rl_index* idx) { return idx->dim > 1024 ? 5 : GEMM_MIN_QUERIES; }

// ---- a5 ----------------------------------------------------------------------------------------------
int rl_adapter_apply(const float* A, const float* queries, int32_t n_queries, int32_t dim, float* out_f32,
                     uint16_t* out_f16, int mem, void* stream) {
    if (dim <= 0 || n_queries < 0) return fail(RL_ERR_INVALID, "rl_adapter_apply: bad size");
    if (!A || (n_queries > 0 && !queries)) return fail(RL_ERR_INVALID, "rl_adapter_apply: null input");
    if (!out_f32 && !out_f16) return fail(RL_ERR_INVALID, "rl_adapter_apply: no output requested");
    if (n_queries == 0) return RL_OK;
    hipStream_t s = as_stream(stream);
    DevBuf t_a, t_q, t_o32, t_o16, t_tmp;
    const float* d_a; const float* d_q;
    RL_TRY(stage_in(A, (size_t)dim * dim, mem, s, t_a, &d_a));
    RL_TRY(stage_in(queries, (size_t)n_queries * dim, mem, s, t_q, &d_q));
    float* d_o32; uint16_t* d_o16;
    RL_TRY(stage_out_begin(out_f32, (size_t)n_queries * dim, mem, t_o32, &d_o32));
    RL_TRY(stage_out_begin(out_f16, (size_t)n_queries * dim, mem, t_o16, &d_o16));
    float* d_res = d_o32;
    if (!d_res) {  // fp16-only output still needs the fp32 result first
        RL_TRY(t_tmp.alloc((size_t)n_queries * dim * sizeof(float)));
        d_res = t_tmp.as<float>();
    }
    // out[b][r] = sum_c A[r][c] q[b][c]: a similarity scan over the rows of A in raw-dot mode.  Batches go
    // through the MFMA tile kernel (32 queries per pass over A, which stays L2-resident), single queries and
    // other dims through the VALU scan.
    bool done = false;
    int dev = 0, n_cu = 256;
    RL_HIP(hipGetDevice(&dev));
    RL_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_queries >= GEMM_MIN_QUERIES) {  // cfg 5's B = 1000: one fp32 MFMA GEMM (score_gemm.hip)
        const int st = launch_score_gemm(d_a, dim, dim, d_q, n_queries, d_res, dim, nullptr, nullptr, nullptr,
                                         SCAN_RAW_DOT, n_cu, s);
        if (st == RL_OK) done = true; else if (st != RL_ERR_UNSUPPORTED) return st;
    }
    if (!done && n_queries > 4) {
        done = true;
        for (int32_t b0 = 0; b0 < n_queries && done; b0 += 32) {
            const int32_t nq = std::min<int32_t>(32, n_queries - b0);
            const int st = launch_maxsim_stream(d_a, dim, dim, d_q + (int64_t)b0 * dim, nq, nullptr, nullptr, 0, 1,
                                                d_res + (int64_t)b0 * dim, dim, n_cu, s);
            if (st == RL_ERR_UNSUPPORTED) done = false; else RL_TRY(st);
        }
    }
    if (!done) RL_TRY(launch_scan_rows(d_a, dim, dim, d_q, n_queries, nullptr, SCAN_RAW_DOT, d_res, dim, s));
    if (d_o16) RL_TRY(launch_cast_f16(d_res, d_o16, (int64_t)n_queries * dim, s));
    RL_TRY(stage_out_end(out_f32, (size_t)n_queries * dim, mem, s, t_o32));
    RL_TRY(stage_out_end(out_f16, (size_t)n_queries * dim, mem, s, t_o16));
    if (mem == RL_MEM_DEVICE && !out_f32) RL_HIP(hipStreamSynchronize(s));  // t_tmp dies with this frame
    return finish(mem, s);
}

// ---- index ---------------------------------------------------------------------------------------------
int rl_index_destroy(rl_index* idx) {
    if (!idx) return RL_OK;
    if (idx->owns_E && idx->E) (void)hipFree(const_cast<float*>(idx->E));
    if (idx->owns_E && idx->E16) (void)hipFree(const_cast<uint16_t*>(idx->E16));
    if (idx->offsets) (void)hipFree(idx->offsets);
    if (idx->row_to_chunk) (void)hipFree(idx->row_to_chunk);
    if (idx->norm) (void)hipFree(idx->norm);
    if (idx->sumsq) (void)hipFree(idx->sumsq);
    if (idx->d_range) (void)hipFree(idx->d_range);
    if (idx->d_norms) (void)hipFree(idx->d_norms);
    if (idx->live_chunk_bits) (void)hipFree(idx->live_chunk_bits);
    if (idx->live_row_bits) (void)hipFree(idx->live_row_bits);
    idx->maskbuf.release();
    idx->qsplit.release();
    idx->planes.release();
    idx->ends.release();
    idx->qplanes.release();
    idx->q32.release();
    if (idx->h_fell_back) (void)hipHostFree(idx->h_fell_back);
    idx->cand.release();
    idx->fused.release();
    idx->pp_work.release();
    idx->rankbuf.release();
    idx->hiplane.release();
    idx->hibuf.release();
    idx->hi_image.release();
    select_workspace_free(idx->ws);
    idx->scores.release();
    idx->hits.release();
    idx->misc.release();
    delete idx;
    return RL_OK;
}

static int index_create_any(rl_index** out, const void* embeddings, bool f16, int64_t n_rows, int32_t dim,
                            const int64_t* chunk_offsets, int64_t n_chunks, int metric, int mem, void* stream) {
    if (!out) return fail(RL_ERR_INVALID, "rl_index_create: null output handle");
    *out = nullptr;
    if (n_rows < 0 || dim <= 0) return fail(RL_ERR_INVALID, "rl_index_create: bad shape");
    if (n_rows > 0 && !embeddings) return fail(RL_ERR_INVALID, "rl_index_create: null embeddings");
    if (n_rows >= (int64_t)0x7fffffff - 1) return fail(RL_ERR_UNSUPPORTED, "rl_index_create: more than 2^31-2 rows");
    if (scan_mode(metric) < 0) return fail(RL_ERR_INVALID, "rl_index_create: unknown metric");
    if (dim > 4096) return fail(RL_ERR_UNSUPPORTED, "rl_index_create: dim must be <= 4096");
    // (the stream kernels' dims; wider -- round 6 -- through the packed scan, the sixteen-query pass and the wide re-scoring kernel)
    if (f16 && dim != 128 && dim != 256 && dim != 384 && dim != 512 && dim != 768 && dim != 1024 && !(dim > 1024 && hi_dim_ok(dim)))
        return fail(RL_ERR_UNSUPPORTED, "rl_index_create_f16: dim must be one of 128, 256, 384, 512, 768, 1024, or a multiple of 128 up to 4096");
    if (f16 && mem == RL_MEM_DEVICE && (reinterpret_cast<uintptr_t>(embeddings) & 15))
        return fail(RL_ERR_INVALID, "rl_index_create_f16: device embeddings must be 16-byte aligned");
    std::vector<int64_t> host_offsets;
    bool has_empty = false;
    if (chunk_offsets) {
        if (n_chunks < 0) return fail(RL_ERR_INVALID, "rl_index_create: negative n_chunks");
        if (chunk_offsets[0] != 0 || chunk_offsets[n_chunks] != n_rows)
            return fail(RL_ERR_INVALID, "rl_index_create: chunk_offsets must start at 0 and end at n_rows");
        for (int64_t c = 0; c < n_chunks; ++c) {
            if (chunk_offsets[c + 1] < chunk_offsets[c])
                return fail(RL_ERR_INVALID, "rl_index_create: chunk_offsets must be ascending");
            has_empty |= chunk_offsets[c + 1] == chunk_offsets[c];
        }
    } else {
        n_chunks = n_rows;
        host_offsets.resize((size_t)n_rows + 1);
        for (int64_t i = 0; i <= n_rows; ++i) host_offsets[(size_t)i] = i;
        chunk_offsets = host_offsets.data();
    }
    hipStream_t s = as_stream(stream);
    rl_index* idx = new rl_index();
    {   // route options: the process-wide defaults as they are now
        std::lock_guard<std::mutex> lock(g_default_opts_mu);
        idx->opt = g_default_opts;
    }
    idx->arithmetic = (int)idx->opt.v[RL_OPT_ARITHMETIC];
    idx->n_rows = n_rows;
    idx->dim = dim;
    idx->n_chunks = n_chunks;
    idx->metric = metric;
    idx->has_empty_chunk = has_empty;
    idx->h_offsets.assign(chunk_offsets, chunk_offsets + n_chunks + 1);
    idx->cap_rows = n_rows;
    idx->cap_chunks = n_chunks;
    auto bail = [&](int code) { rl_index_destroy(idx); return code; };
#define RL_IDX(expr) do { int _s = (expr); if (_s != RL_OK) return bail(_s); } while (0)
#define RL_IDX_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(_e == hipErrorOutOfMemory ? RL_ERR_NOMEM : RL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); } while (0)
    int dev = 0;
    RL_IDX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RL_IDX_HIP(hipGetDeviceProperties(&prop, dev));
    idx->n_cu = prop.multiProcessorCount;
    const size_t elt = f16 ? sizeof(uint16_t) : sizeof(float);
    const void* dev_rows = embeddings;
    if (mem == RL_MEM_HOST) {
        void* d = nullptr;
        RL_IDX_HIP(hipMalloc(&d, std::max<size_t>((size_t)n_rows * dim * elt, 16)));
        idx->owns_E = true;
        dev_rows = d;
        (f16 ? (const void*&)idx->E16 : (const void*&)idx->E) = d;  // owned from here on (bail() frees it)
        if (n_rows) RL_IDX_HIP(hipMemcpyAsync(d, embeddings, (size_t)n_rows * dim * elt, hipMemcpyHostToDevice, s));
    }
    if (f16) idx->E16 = static_cast<const uint16_t*>(dev_rows); else idx->E = static_cast<const float*>(dev_rows);
    RL_IDX_HIP(hipMalloc(&idx->offsets, (size_t)(n_chunks + 1) * sizeof(int64_t)));
    RL_IDX_HIP(hipMemcpyAsync(idx->offsets, chunk_offsets, (size_t)(n_chunks + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    RL_IDX_HIP(hipMalloc(&idx->row_to_chunk, (size_t)(n_rows + 65) * sizeof(int32_t)));  // +1 terminator, +64 pad
    RL_IDX(launch_row_to_chunk(idx->offsets, n_chunks, n_rows, idx->row_to_chunk, s));
    if (metric == RL_COSINE) RL_IDX_HIP(hipMalloc(&idx->norm, (size_t)n_rows * sizeof(float) + 64));  // (+ 64 B: a 16-row block's norms are readable with one scalar load)
    if (metric == RL_L2) RL_IDX_HIP(hipMalloc(&idx->sumsq, std::max<size_t>((size_t)n_rows * sizeof(float), 16)));
    if (idx->norm || idx->sumsq)
        RL_IDX(f16 ? launch_row_norms16(idx->E16, n_rows, dim, idx->norm, idx->sumsq, s)
                   : launch_row_norms(idx->E, n_rows, dim, idx->norm, idx->sumsq, s));
    RL_IDX(scan_row_range(idx, 0, n_rows, s));
    RL_IDX(refresh_planes(idx, s));
    RL_IDX(refresh_hi_plane(idx, s));
    RL_IDX_HIP(hipStreamSynchronize(s));  // host_offsets / caller buffers may go away after return
#undef RL_IDX
#undef RL_IDX_HIP
    *out = idx;
    return RL_OK;
}

int rl_index_create(rl_index** out, const float* embeddings, int64_t n_rows, int32_t dim,
                    const int64_t* chunk_offsets, int64_t n_chunks, int metric, int mem, void* stream) {
    return index_create_any(out, embeddings, false, n_rows, dim, chunk_offsets, n_chunks, metric, mem, stream);
}

int rl_index_create_f16(rl_index** out, const uint16_t* embeddings_f16, int64_t n_rows, int32_t dim,
                        const int64_t* chunk_offsets, int64_t n_chunks, int metric, int mem, void* stream) {
    return index_create_any(out, embeddings_f16, true, n_rows, dim, chunk_offsets, n_chunks, metric, mem, stream);
}

// ---- lifecycle: append / delete -------------------------------------------------
namespace {
// (Re)build the device live bitsets from idx->h_live.
int upload_live_bits(rl_index* idx, hipStream_t s) {
    if (idx->h_live.empty()) return RL_OK;
    const size_t cw = (size_t)(idx->n_chunks + 31) / 32, rw = (size_t)(idx->n_rows + 31) / 32;
    if (idx->live_chunk_bits) (void)hipFree(idx->live_chunk_bits);
    if (idx->live_row_bits) (void)hipFree(idx->live_row_bits);
    idx->live_chunk_bits = idx->live_row_bits = nullptr;
    RL_HIP(hipMalloc(&idx->live_chunk_bits, std::max<size_t>(cw * 4, 16)));
    RL_HIP(hipMalloc(&idx->live_row_bits, std::max<size_t>(rw * 4, 16)));
    RL_HIP(hipMemcpyAsync(idx->live_chunk_bits, idx->h_live.data(), cw * 4, hipMemcpyHostToDevice, s));
    RL_TRY(launch_expand_chunk_bits(idx->live_chunk_bits, idx->row_to_chunk, idx->n_rows, nullptr, idx->live_row_bits, s));
    RL_HIP(hipStreamSynchronize(s));
    return RL_OK;
}
}  // namespace

int rl_index_delete_chunks(rl_index* idx, const int64_t* chunk_ordinals, int64_t n, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_delete_chunks: null index");
    if (n < 0 || (n > 0 && !chunk_ordinals)) return fail(RL_ERR_INVALID, "rl_index_delete_chunks: bad arguments");
    if (n == 0) return RL_OK;
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    for (int64_t i = 0; i < n; ++i)
        if (chunk_ordinals[i] < 0 || chunk_ordinals[i] >= idx->n_chunks)
            return fail(RL_ERR_INVALID, "rl_index_delete_chunks: chunk ordinal out of range");
    RL_TRY(use_scratch(idx, s));  // the live bitsets are read by searches that may still run on another stream
    const size_t cw = (size_t)(idx->n_chunks + 31) / 32;
    if (idx->h_live.empty()) idx->h_live.assign(cw, 0xffffffffu);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t c = chunk_ordinals[i];
        uint32_t& w = idx->h_live[(size_t)(c >> 5)];
        const uint32_t bit = 1u << (c & 31);
        if (w & bit) {
            w &= ~bit;
            ++idx->n_dead_chunks;
            idx->n_dead_rows += idx->h_offsets[(size_t)c + 1] - idx->h_offsets[(size_t)c];
        }
    }
    return upload_live_bits(idx, s);
}

int rl_index_live(rl_index* idx, int64_t* live_rows, int64_t* live_chunks, void* stream) {
    (void)stream;
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_live: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (live_rows) *live_rows = idx->n_rows - idx->n_dead_rows;
    if (live_chunks) *live_chunks = idx->n_chunks - idx->n_dead_chunks;
    return RL_OK;
}

int rl_index_filter_stats(rl_index* idx, int64_t out[6], void* stream) {
    if (!idx || !out) return fail(RL_ERR_INVALID, "rl_index_filter_stats: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    const auto f = idx->filt;  // (use_scratch forgets the records: every other call may re-reserve the pools they point into; this one does not)
    const bool replay_valid = idx->replay.valid;
    RL_TRY(use_scratch(idx, s));
    idx->filt = f;
    idx->replay.valid = replay_valid;
    for (int i = 0; i < 6; ++i) out[i] = 0;
    if (f.kind == RL_FILTER_NONE || f.n <= 0) return RL_OK;
    RL_HIP(hipStreamSynchronize(s));
    std::vector<uint32_t> cnt((size_t)f.n);
    uint32_t flag = 0;
    RL_HIP(hipMemcpy(cnt.data(), f.cnt, (size_t)f.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    RL_HIP(hipMemcpy(&flag, f.flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
    int64_t sum = 0, mx = 0;
    for (uint32_t c : cnt) { sum += c; mx = std::max<int64_t>(mx, c); }
    out[0] = f.kind; out[1] = f.n; out[2] = sum; out[3] = mx; out[4] = f.cap; out[5] = flag != 0;
    return RL_OK;
}

int rl_index_compact(rl_index* idx, int64_t* out_remap, int64_t* new_n_rows, int64_t* new_n_chunks, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_compact: null index");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    const int64_t old_c = idx->n_chunks;
    auto identity = [&]() {
        if (out_remap) for (int64_t c = 0; c < old_c; ++c) out_remap[c] = c;
        if (new_n_rows) *new_n_rows = idx->n_rows;
        if (new_n_chunks) *new_n_chunks = idx->n_chunks;
        return RL_OK;
    };
    if (idx->n_dead_chunks == 0) return identity();
    // ---- the surviving chunks, in their old order: new CSR, old ordinal -> new ordinal, new row -> old row ---------------
    std::vector<int64_t> new_off, remap((size_t)old_c, -1), old_row;
    new_off.reserve((size_t)(old_c - idx->n_dead_chunks) + 1);
    old_row.reserve((size_t)(idx->n_rows - idx->n_dead_rows));
    new_off.push_back(0);
    bool has_empty = false;
    for (int64_t c = 0; c < old_c; ++c) {
        if (!((idx->h_live[(size_t)(c >> 5)] >> (c & 31)) & 1u)) continue;
        remap[(size_t)c] = (int64_t)new_off.size() - 1;
        for (int64_t r = idx->h_offsets[(size_t)c]; r < idx->h_offsets[(size_t)c + 1]; ++r) old_row.push_back(r);
        has_empty |= idx->h_offsets[(size_t)c + 1] == idx->h_offsets[(size_t)c];
        new_off.push_back((int64_t)old_row.size());
    }
    const int64_t new_n = (int64_t)old_row.size(), new_c = (int64_t)new_off.size() - 1;
    const bool f16 = idx->E16 != nullptr;
    const size_t row_bytes = (size_t)idx->dim * (f16 ? sizeof(uint16_t) : sizeof(float));
    if (row_bytes % 4) return fail(RL_ERR_UNSUPPORTED, "rl_index_compact: an odd dim of fp16 rows is not supported");
    // ---- all-or-nothing: every new buffer exists before the index is touched ------------------------------------------------
    void *e = nullptr, *map = nullptr;
    float *nn = nullptr, *ns = nullptr;
    int32_t* r2c = nullptr;
    int64_t* o = nullptr;
    auto undo = [&](int code) {
        for (void* p : {e, map, (void*)nn, (void*)ns, (void*)r2c, (void*)o}) if (p) (void)hipFree(p);
        return code;
    };
#define RL_CMP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return undo(fail(_e == hipErrorOutOfMemory ? RL_ERR_NOMEM : RL_ERR_HIP, std::string("rl_index_compact: ") + hipGetErrorString(_e))); } while (0)
    RL_CMP(hipMalloc(&e, std::max<size_t>((size_t)new_n * row_bytes, 16)));
    RL_CMP(hipMalloc(&map, std::max<size_t>((size_t)new_n * sizeof(int64_t), 16)));
    if (idx->norm) RL_CMP(hipMalloc(&nn, (size_t)new_n * sizeof(float) + 64));
    if (idx->sumsq) RL_CMP(hipMalloc(&ns, std::max<size_t>((size_t)new_n * sizeof(float), 16)));
    RL_CMP(hipMalloc(&r2c, (size_t)(new_n + 65) * sizeof(int32_t)));
    RL_CMP(hipMalloc(&o, (size_t)(new_c + 1) * sizeof(int64_t)));
    if (new_n) RL_CMP(hipMemcpyAsync(map, old_row.data(), (size_t)new_n * sizeof(int64_t), hipMemcpyHostToDevice, s));
    RL_CMP(hipMemcpyAsync(o, new_off.data(), (size_t)(new_c + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    {
        const void* src = f16 ? (const void*)idx->E16 : (const void*)idx->E;
        const int st = launch_compact_rows(src, (int64_t)row_bytes, static_cast<const int64_t*>(map), new_n, e, s);
        if (st != RL_OK) return undo(st);
        if (nn) { const int st2 = launch_compact_rows(idx->norm, 4, static_cast<const int64_t*>(map), new_n, nn, s); if (st2 != RL_OK) return undo(st2); }
        if (ns) { const int st2 = launch_compact_rows(idx->sumsq, 4, static_cast<const int64_t*>(map), new_n, ns, s); if (st2 != RL_OK) return undo(st2); }
    }
    RL_CMP(hipStreamSynchronize(s));
#undef RL_CMP
    // ---- commit -----------------------------------------------------------------------------------------------------------------
    if (idx->owns_E) (void)hipFree(const_cast<void*>(f16 ? (const void*)idx->E16 : (const void*)idx->E));
    if (f16) idx->E16 = static_cast<const uint16_t*>(e); else idx->E = static_cast<const float*>(e);
    idx->owns_E = true;
    (void)hipFree(map);
    if (idx->norm) { (void)hipFree(idx->norm); idx->norm = nn; }
    if (idx->sumsq) { (void)hipFree(idx->sumsq); idx->sumsq = ns; }
    (void)hipFree(idx->row_to_chunk);
    idx->row_to_chunk = r2c;
    (void)hipFree(idx->offsets);
    idx->offsets = o;
    idx->n_rows = idx->cap_rows = new_n;
    idx->n_chunks = idx->cap_chunks = new_c;
    idx->h_offsets = std::move(new_off);
    idx->has_empty_chunk = has_empty;
    idx->h_live.clear();
    if (idx->live_chunk_bits) { (void)hipFree(idx->live_chunk_bits); idx->live_chunk_bits = nullptr; }
    if (idx->live_row_bits) { (void)hipFree(idx->live_row_bits); idx->live_row_bits = nullptr; }
    idx->n_dead_chunks = idx->n_dead_rows = 0;
    RL_TRY(launch_row_to_chunk(idx->offsets, new_c, new_n, idx->row_to_chunk, s));
    // the magnitude range may only have shrunk: recompute it over the survivors, then the corpus image from scratch
    idx->max_abs = 0.f;
    idx->min_row_max = std::numeric_limits<float>::infinity();
    idx->nonfinite = false;
    idx->planes_rows = 0;
    idx->planes_scale = 0.f;
    idx->hi_rows = 0;
    idx->hi_scale = 0.f;
    idx->hi_image_rows = 0;
    idx->hi_image_scale = 0.f;
    idx->max_row_norm = idx->max_lo_norm = idx->max_lo_ratio = 0.f;
    idx->min_row_norm = std::numeric_limits<float>::infinity();
    idx->max_row_norm_rows = 0;
    RL_TRY(scan_row_range(idx, 0, new_n, s));
    RL_TRY(refresh_planes(idx, s));
    RL_TRY(refresh_hi_plane(idx, s));
    RL_HIP(hipStreamSynchronize(s));
    if (out_remap) std::memcpy(out_remap, remap.data(), (size_t)old_c * sizeof(int64_t));
    if (new_n_rows) *new_n_rows = new_n;
    if (new_n_chunks) *new_n_chunks = new_c;
    return RL_OK;
}

int rl_index_set_arithmetic(rl_index* idx, int mode) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_set_arithmetic: null index");
    if (mode != RL_ARITH_AUTO && mode != RL_ARITH_FP32_EXACT) return fail(RL_ERR_INVALID, "rl_index_set_arithmetic: unknown mode");
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, nullptr));
    idx->arithmetic = mode;
    idx->opt.v[RL_OPT_ARITHMETIC] = mode;
    update_split_scale(idx);
    RL_TRY(refresh_planes(idx, nullptr));
    RL_TRY(refresh_hi_plane(idx, nullptr));
    RL_HIP(hipStreamSynchronize(nullptr));
    return RL_OK;
}

int rl_set_default_option(int key, int64_t value) {
    if (!option_value_ok(key, value)) return fail(RL_ERR_INVALID, "rl_set_default_option: unknown key or value out of range");
    std::lock_guard<std::mutex> lock(g_default_opts_mu);
    g_default_opts.v[key] = value;
    return RL_OK;
}

int rl_get_default_option(int key, int64_t* value) {
    if (!value || key < 1 || key >= RL_OPT_COUNT_) return fail(RL_ERR_INVALID, "rl_get_default_option: unknown key");
    std::lock_guard<std::mutex> lock(g_default_opts_mu);
    *value = g_default_opts.v[key];
    return RL_OK;
}

int rl_index_set_option(rl_index* idx, int key, int64_t value) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_set_option: null index");
    if (!option_value_ok(key, value)) return fail(RL_ERR_INVALID, "rl_index_set_option: unknown key or value out of range");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (idx->opt.v[key] == value) return RL_OK;
    idx->opt.v[key] = value;
    if (key == RL_OPT_KEEP_IMAGE || key == RL_OPT_KEEP_HI || key == RL_OPT_KEEP_HI_PLANE || key == RL_OPT_IMAGE_HEADROOM_MB || key == RL_OPT_ARITHMETIC ||
        key == RL_OPT_LAZY_IMAGES) {
        // what the index keeps in device memory changes: rebuild / release now (synchronous, like rl_index_set_arithmetic)
        RL_TRY(use_scratch(idx, nullptr));
        if (key == RL_OPT_ARITHMETIC) {
            idx->arithmetic = (int)value;
            update_split_scale(idx);
        }
        RL_TRY(refresh_planes(idx, nullptr));
        RL_TRY(refresh_hi_plane(idx, nullptr));
        RL_HIP(hipStreamSynchronize(nullptr));
    }
    return RL_OK;
}

int rl_index_get_option(rl_index* idx, int key, int64_t* value) {
    if (!idx || !value || key < 1 || key >= RL_OPT_COUNT_) return fail(RL_ERR_INVALID, "rl_index_get_option: null argument or unknown key");
    std::lock_guard<std::mutex> lock(idx->mu);
    *value = idx->opt.v[key];
    return RL_OK;
}

int rl_index_arithmetic(rl_index* idx, int* in_effect) {
    if (!idx || !in_effect) return fail(RL_ERR_INVALID, "rl_index_arithmetic: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *in_effect = idx->E16 ? RL_ARITH_F16_STORED : (idx->split_scale > 0.f ? RL_ARITH_F16_SPLIT : RL_ARITH_FP32_EXACT);
    return RL_OK;
}

int rl_index_append(rl_index* idx, const float* rows, int64_t n_new_rows, const int64_t* new_chunk_sizes,
                    int64_t n_new_chunks, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_append: null index");
    if (n_new_rows < 0 || n_new_chunks < 0) return fail(RL_ERR_INVALID, "rl_index_append: negative size");
    if (!new_chunk_sizes) n_new_chunks = n_new_rows;
    if (n_new_rows == 0 && n_new_chunks == 0) return RL_OK;
    if (n_new_rows > 0 && !rows) return fail(RL_ERR_INVALID, "rl_index_append: null rows");
    if (new_chunk_sizes) {
        int64_t tot = 0;
        for (int64_t c = 0; c < n_new_chunks; ++c) {
            if (new_chunk_sizes[c] < 0) return fail(RL_ERR_INVALID, "rl_index_append: negative chunk size");
            tot += new_chunk_sizes[c];
        }
        if (tot != n_new_rows) return fail(RL_ERR_INVALID, "rl_index_append: chunk sizes must sum to n_new_rows");
    }
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));  // storage may be reallocated under searches still running on another stream
    const int64_t old_n = idx->n_rows, new_n = old_n + n_new_rows;
    const int64_t old_c = idx->n_chunks, new_c = old_c + n_new_chunks;
    if (new_n >= (int64_t)0x7fffffff - 1) return fail(RL_ERR_UNSUPPORTED, "rl_index_append: more than 2^31-2 rows");
    const bool f16 = idx->E16 != nullptr;
    const size_t row_bytes = (size_t)idx->dim * (f16 ? sizeof(uint16_t) : sizeof(float));
    const void* old_rows = f16 ? (const void*)idx->E16 : (const void*)idx->E;
    // ---- storage: own it, grow geometrically.  All-or-nothing: every new buffer is allocated (and filled) before the
    // index is touched, so a failed allocation leaves the index exactly as it was. -----------------------------------
    if (!idx->owns_E || new_n > idx->cap_rows || new_c > idx->cap_chunks) {
        const bool grow_rows = !idx->owns_E || new_n > idx->cap_rows;
        const bool grow_chunks = new_c > idx->cap_chunks;
        const int64_t cap = grow_rows ? std::max<int64_t>(new_n, idx->cap_rows + idx->cap_rows / 2) : idx->cap_rows;
        const int64_t ccap = grow_chunks ? std::max<int64_t>(new_c, idx->cap_chunks + idx->cap_chunks / 2) : idx->cap_chunks;
        void* e = nullptr;
        float *nn = nullptr, *ns = nullptr;
        int32_t* r2c = nullptr;
        int64_t* o = nullptr;
        auto undo = [&](int code) {
            for (void* p : {e, (void*)nn, (void*)ns, (void*)r2c, (void*)o}) if (p) (void)hipFree(p);
            return code;
        };
#define RL_GROW(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return undo(fail(_e == hipErrorOutOfMemory ? RL_ERR_NOMEM : RL_ERR_HIP, std::string("rl_index_append: ") + hipGetErrorString(_e))); } while (0)
        if (grow_rows) {
            RL_GROW(hipMalloc(&e, std::max<size_t>((size_t)cap * row_bytes, 16)));
            if (idx->norm) RL_GROW(hipMalloc(&nn, (size_t)cap * sizeof(float) + 64));
            if (idx->sumsq) RL_GROW(hipMalloc(&ns, std::max<size_t>((size_t)cap * sizeof(float), 16)));
            RL_GROW(hipMalloc(&r2c, (size_t)(cap + 65) * sizeof(int32_t)));
        }
        if (grow_chunks) RL_GROW(hipMalloc(&o, (size_t)(ccap + 1) * sizeof(int64_t)));
        if (grow_rows && old_n) {
            RL_GROW(hipMemcpyAsync(e, old_rows, (size_t)old_n * row_bytes, hipMemcpyDeviceToDevice, s));
            if (nn) RL_GROW(hipMemcpyAsync(nn, idx->norm, (size_t)old_n * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (ns) RL_GROW(hipMemcpyAsync(ns, idx->sumsq, (size_t)old_n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        RL_GROW(hipStreamSynchronize(s));
#undef RL_GROW
        // commit
        if (grow_rows) {
            if (idx->owns_E && old_rows) (void)hipFree(const_cast<void*>(old_rows));
            if (idx->norm) { (void)hipFree(idx->norm); idx->norm = nn; }
            if (idx->sumsq) { (void)hipFree(idx->sumsq); idx->sumsq = ns; }
            (void)hipFree(idx->row_to_chunk);
            if (f16) idx->E16 = static_cast<const uint16_t*>(e); else idx->E = static_cast<const float*>(e);
            idx->owns_E = true;
            idx->row_to_chunk = r2c;
            idx->cap_rows = cap;
        }
        if (grow_chunks) {
            (void)hipFree(idx->offsets);
            idx->offsets = o;
            idx->cap_chunks = ccap;
        }
    }
    // ---- new rows, CSR, ordinals, norms ------------------------------------------------------------------
    DevBuf t_rows;
    if (n_new_rows && !f16)
        RL_HIP(hipMemcpyAsync(const_cast<float*>(idx->E) + (size_t)old_n * idx->dim, rows, (size_t)n_new_rows * row_bytes,
                              mem == RL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    if (n_new_rows && f16) {  // fp32 rows in, rounded to nearest even into the fp16 store (the reference's astype(float16))
        const float* d_rows;
        RL_TRY(stage_in(rows, (size_t)n_new_rows * idx->dim, mem, s, t_rows, &d_rows));
        RL_TRY(launch_cast_f16(d_rows, const_cast<uint16_t*>(idx->E16) + (size_t)old_n * idx->dim,
                               n_new_rows * (int64_t)idx->dim, s));
    }
    idx->h_offsets.reserve((size_t)new_c + 1);
    for (int64_t c = 0; c < n_new_chunks; ++c) {
        const int64_t sz = new_chunk_sizes ? new_chunk_sizes[c] : 1;
        idx->has_empty_chunk |= sz == 0;
        idx->h_offsets.push_back(idx->h_offsets.back() + sz);
    }
    idx->n_rows = new_n;
    idx->n_chunks = new_c;
    RL_HIP(hipMemcpyAsync(idx->offsets, idx->h_offsets.data(), (size_t)(new_c + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    RL_TRY(launch_row_to_chunk(idx->offsets, new_c, new_n, idx->row_to_chunk, s));
    if ((idx->norm || idx->sumsq) && n_new_rows) {
        float* nn = idx->norm ? idx->norm + old_n : nullptr;
        float* ns = idx->sumsq ? idx->sumsq + old_n : nullptr;
        RL_TRY(f16 ? launch_row_norms16(idx->E16 + (size_t)old_n * idx->dim, n_new_rows, idx->dim, nn, ns, s)
                   : launch_row_norms(idx->E + (size_t)old_n * idx->dim, n_new_rows, idx->dim, nn, ns, s));
    }
    RL_TRY(scan_row_range(idx, old_n, n_new_rows, s));
    RL_TRY(refresh_planes(idx, s));
    RL_TRY(refresh_hi_plane(idx, s));
    if (!idx->h_live.empty()) {  // new chunks are live
        const size_t cw = (size_t)(new_c + 31) / 32;
        idx->h_live.resize(cw, 0u);
        for (int64_t c = old_c; c < new_c; ++c) idx->h_live[(size_t)(c >> 5)] |= 1u << (c & 31);
        RL_TRY(upload_live_bits(idx, s));
    }
    return sync_and_drain(s);  // the caller's buffers may go away after return
}

int rl_index_info(const rl_index* idx, int64_t* n_rows, int32_t* dim, int64_t* n_chunks, int* metric) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_info: null index");
    if (n_rows) *n_rows = idx->n_rows;
    if (dim) *dim = idx->dim;
    if (n_chunks) *n_chunks = idx->n_chunks;
    if (metric) *metric = idx->metric;
    return RL_OK;
}

int rl_index_prepare(rl_index* idx, uint32_t images, uint32_t* built, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_index_prepare: null index");
    if (images & ~(uint32_t)IMG_ALL) return fail(RL_ERR_INVALID, "rl_index_prepare: unknown image bit");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    idx->no_room_retry_epoch = 0;  // (an explicit request always tries)
    RL_TRY(demand_images(idx, images, s));
    if (built) *built = (image_valid(idx) ? IMG_PLANES : 0u) | (hi_image_valid(idx) ? IMG_HI_IMAGE : 0u) | (hi_valid(idx) ? IMG_HI_PLANE : 0u);
    return RL_OK;
}

int rl_index_memory(const rl_index* idx, int64_t out[8]) {
    if (!idx || !out) return fail(RL_ERR_INVALID, "rl_index_memory: null argument");
    out[0] = (int64_t)idx->n_rows * idx->dim * (idx->E16 ? 2 : 4);
    out[1] = image_valid(idx) ? (int64_t)idx->planes.cap : 0;
    out[2] = hi_image_valid(idx) ? (int64_t)idx->hi_image.cap : 0;
    out[3] = hi_valid(idx) ? (int64_t)idx->hiplane.cap : 0;
    out[4] = (int64_t)(idx->scores.cap + idx->hits.cap + idx->misc.cap + idx->maskbuf.cap + idx->qsplit.cap + idx->qplanes.cap + idx->q32.cap + idx->cand.cap +
                       idx->fused.cap + idx->pp_work.cap + idx->rankbuf.cap + idx->hibuf.cap + idx->ends.cap);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = total_b = 0; }
    out[5] = (int64_t)free_b;
    out[6] = (int64_t)total_b;
    out[7] = (int64_t)image_headroom(idx);
    return RL_OK;
}

// ---- a6 + a7 -----------------------------------------------------------------------------------------
namespace {

// Similarity of `nb` device queries against every row -> idx->scores [nb x ld] (device).
// hist_done != nullptr: the caller ranks the scores next and there is no row mask in between -- where the path ends in the
// metric transform of raw dots, the selection's histogram is taken in the same launch and *hist_done says so.
int score_rows(rl_index* idx, const float* d_q, int32_t nb, int64_t ld, hipStream_t s, bool* hist_done = nullptr) {
    auto transform = [&](float* sc_, int mode_) {
        if (hist_done) {
            *hist_done = true;
            return launch_transform_hist(sc_, nb, idx->n_rows, ld, idx->norm, idx->sumsq, d_q, idx->dim, mode_, idx->ws, s);
        }
        return launch_transform(sc_, nb, idx->n_rows, ld, idx->norm, idx->sumsq, d_q, idx->dim, mode_, s);
    };
    const int mode = scan_mode(idx->metric);
    float* sc = idx->scores.as<float>();
    if (nb >= rows_gemm_min(idx) && idx->opt.on(RL_OPT_PLANES_GEMM)) RL_TRY(demand_images(idx, IMG_PLANES, s));
    if (nb >= rows_gemm_min(idx) && image_valid(idx)) {
        // the row-score GEMM over the corpus image (maxsim_gemm.hip MODE 1; fp32 corpus: pre-split planes, fp16-stored corpus:
        // its one-plane image): no conversion in the loop
        if (idx->opt.on(RL_OPT_PLANES_GEMM)) {
            RL_TRY(idx->misc.reserve(score_planes_scratch_floats(nb, idx->dim) * sizeof(float)));
            const int st = launch_score_planes(idx->planes.p, idx->n_rows, idx->dim, d_q, nb, sc, ld, idx->norm, idx->sumsq,
                                               idx->misc.as<float>(), mode, idx->n_cu, s, image_scale(idx), idx->E16 != nullptr);
            if (st != RL_ERR_UNSUPPORTED) return st;
        }
    }
    if (idx->E16) {  // fp16 storage: f16-MFMA stream passes of up to 32 queries, whatever the batch size: the VALU scan
        // (scan16.hip) was measured at 0.37 ms per 1 M x 1024 pass against 0.31 ms for one stream pass + transform -- with
        // half the bytes per row the LDS-DMA stream is the faster reader even for a single query.  Exception: l2 with up
        // to 4 queries keeps the scan, which sums (e - q)^2 directly; the stream path's |e|^2 + |q|^2 - 2 e.q loses a
        // near-duplicate's small distance to cancellation (the reference's nearest neighbour IS often a near-duplicate).
        // A WIDE fp16-stored index (dim > 1024, round 6): the packed scan in every metric, up to four queries per pass -- no stream kernel
        // covers it; batches of five and more took the GEMM over its image above (rows_gemm_min).
        if ((nb <= 4 && mode == SCAN_L2) || idx->dim > 1024)
            return launch_scan_rows16(idx->E16, idx->n_rows, idx->dim, d_q, nb, idx->norm, mode, sc, ld, s);
        for (int32_t b0 = 0; b0 < nb; b0 += 32) {
            const int32_t nq = std::min<int32_t>(32, nb - b0);
            RL_TRY(launch_maxsim_stream16(idx->E16, idx->n_rows, idx->dim, d_q + (int64_t)b0 * idx->dim, nq,
                                          idx->row_to_chunk, idx->offsets, idx->n_chunks, 1, sc + (int64_t)b0 * ld, ld,
                                          idx->n_cu, s));
        }
        return transform(sc, mode);
    }
    if (nb >= rows_gemm_min(idx)) {  // MFMA-bound regime: one 128 x 128-tiled GEMM instead of a corpus pass per 32 queries
        RL_TRY(idx->misc.reserve(score_gemm_scratch_floats(nb, idx->dim, idx->split_scale > 0.f) * sizeof(float)));
        const int st = launch_score_gemm(idx->E, idx->n_rows, idx->dim, d_q, nb, sc, ld, idx->norm, idx->sumsq,
                                         idx->misc.as<float>(), mode, idx->n_cu, s, idx->split_scale);
        if (st != RL_ERR_UNSUPPORTED) return st;
    }
    // Small batches too: one LDS-DMA stream pass (16-query variant: half the MFMAs) + the 4-MB transform reads the
    // corpus at 6.9 TB/s, the VALU scan's global loads at 6.4 (measured, 1 M x 1024: 0.594 vs 0.639 ms).  l2 keeps the
    // scan, which sums (e - q)^2 directly.  launch_maxsim_stream reports RL_ERR_UNSUPPORTED for dims outside its fast
    // path; those fall through to the scan as well.  Below ~256 MB of corpus the scan's lower fixed cost wins
    // (10 k rows: 28 vs 44 us per query).
    const bool big = (int64_t)idx->n_rows * idx->dim >= (int64_t(64) << 20);
    if (nb > 4 || (mode != SCAN_L2 && big)) {
        // MFMA tile kernel, 32 queries per corpus pass, raw dots; then the metric transform.
        bool ok = true;
        for (int32_t b0 = 0; b0 < nb && ok; b0 += 32) {
            const int32_t nq = std::min<int32_t>(32, nb - b0);
            const int st = launch_maxsim_stream(idx->E, idx->n_rows, idx->dim, d_q + (int64_t)b0 * idx->dim, nq,
                                                idx->row_to_chunk, idx->offsets, idx->n_chunks, 1,
                                                sc + (int64_t)b0 * ld, ld, idx->n_cu, s, idx->split_scale);
            if (st == RL_ERR_UNSUPPORTED) ok = false; else RL_TRY(st);
        }
        if (ok) return transform(sc, mode);
    }
    return launch_scan_rows(idx->E, idx->n_rows, idx->dim, d_q, nb, idx->norm, mode, sc, ld, s);
}

// Effective row mask of a call: tombstones and-ed with the expanded chunk filter (nullptr = every row takes part).
int effective_row_mask(rl_index* idx, const uint32_t* d_chunk_filter, hipStream_t s, const uint32_t** out) {
    *out = idx->live_row_bits;
    if (!d_chunk_filter || idx->n_rows == 0) return RL_OK;
    RL_TRY(idx->maskbuf.reserve((size_t)((idx->n_rows + 31) / 32) * sizeof(uint32_t)));
    RL_TRY(launch_expand_chunk_bits(d_chunk_filter, idx->row_to_chunk, idx->n_rows, idx->live_row_bits,
                                    idx->maskbuf.as<uint32_t>(), s));
    *out = idx->maskbuf.as<uint32_t>();
    return RL_OK;
}

// Exact row top-k of a big batch WITHOUT the [B x N] score matrix (BASELINE cfg 5: 1000 queries x 1.25 M rows would write and
// re-read 5 GB twice).  (1) the row-score GEMM over every stride-th 256-row tile of the pre-split image -> a small score
// matrix -> its exact top-k: the k-th best of a subset is a LOWER bound of the k-th best overall, so (2) a second GEMM pass
// over all rows only has to keep the scores that reach their query's bound -- ~k * stride per query, appended to
// per-query candidate lists from the epilogue (maxsim_gemm.hip MODE 2) -- and (3) rl_merge_topk's kernel ranks each list
// exactly, ties to the lowest row.  Same results as the dense path bit for bit (the kept scores are computed by the same
// statements).  A list that overflows (massive ties, an unlucky sample) or an unusable bound sets a device flag, on which
// (4) the dense GEMM + selection run as a guarded fallback -- launched always, returning at once when the flag is clear, so
// nothing here synchronises with the host.  cosine / dot, k <= 512, no row mask; RL_ERR_UNSUPPORTED otherwise.
int search_rows_fused(rl_index* idx, const float* d_q, int32_t B, int32_t k, float* d_scores, int32_t* d_rows, int64_t ld,
                      hipStream_t s) {
    const bool off = !idx->opt.on(RL_OPT_FUSED_TOPK);
    const int cap_env = (int)idx->opt.v[RL_OPT_FUSED_TOPK_CAP];  // tests: force list overflows
    const int mode = scan_mode(idx->metric);
    if (off || B < rows_gemm_min(idx) || k > 512 || (mode != SCAN_COSINE && mode != SCAN_DOT)) return RL_ERR_UNSUPPORTED;
    if (!image_valid(idx)) return RL_ERR_UNSUPPORTED;
    const bool half = idx->E16 != nullptr;
    const float img_scale = image_scale(idx);
    const int64_t n = idx->n_rows, T = (n + 255) / 256;
    const int32_t cap = cap_env > 0 ? std::min(cap_env, MERGE_CAP) : MERGE_CAP;
    // Sample stride: ~k * stride rows per query reach the bound -- a third of a list on average (the fluctuation is
    // ~sqrt(k) * stride) -- and at least 8 sample tiles.  Measured at cfg 5 (same box): stride 27 (this rule) 7.50 ms, 20: 7.67
    // (more sample work), 39 (two pairs per workgroup in the sample pass instead of 2.83): 8.0 -- the candidates' epilogue and
    // list work grows faster than the sample pass shrinks.
    int32_t stride = (int32_t)std::min<int64_t>(MERGE_CAP / (3 * (int64_t)k), T / 8);
    if (idx->opt.v[RL_OPT_FUSED_TOPK_STRIDE] > 0) stride = (int32_t)std::min<int64_t>(idx->opt.v[RL_OPT_FUSED_TOPK_STRIDE], T / 8);  // A/B
    if (stride < 2) return RL_ERR_UNSUPPORTED;
    const int64_t Tv = (T + stride - 1) / stride, ld_s = Tv * 256;
    if (ld_s < k) return RL_ERR_UNSUPPORTED;
    // ---- scratch ------------------------------------------------------------------------------------------------------------
    RL_TRY(idx->misc.reserve(score_planes_scratch_floats(B, idx->dim) * sizeof(float)));
    const size_t n_sample = (size_t)B * ld_s, n_top = (size_t)B * k, n_cand = (size_t)B * cap;
    RL_TRY(idx->fused.reserve((n_sample + 2 * n_top + 2 * n_cand + (size_t)B + 16) * 4));
    float* S_s = idx->fused.as<float>();
    float* top_s = S_s + n_sample;
    int32_t* top_i = reinterpret_cast<int32_t*>(top_s + n_top);
    float* c_s = reinterpret_cast<float*>(top_i + n_top);
    int32_t* c_i = reinterpret_cast<int32_t*>(c_s + n_cand);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(c_i + n_cand);
    uint32_t* flag = cnt + B;
    float* qs = idx->misc.as<float>();
    float* sc = idx->scores.as<float>();  // [B x ld], reserved by the caller: only the fallback touches it
    // ---- (1) sample pass + its exact top-k --------------------------------------------------------------------------------------
    RL_TRY(launch_score_planes_queries(d_q, B, idx->dim, qs, mode, s));
    // (rows past the corpus in the last sampled tile: the pass writes them as -inf itself)
    RL_TRY(launch_score_planes_pass(idx->planes.p, n, idx->dim, B, qs, S_s, ld_s, idx->norm, idx->sumsq, mode, stride, nullptr, nullptr,
                                    idx->n_cu, s, img_scale, half));
    RL_TRY(launch_topk(S_s, B, ld_s, ld_s, k, idx->ws, top_s, top_i, s));
    // ---- (2) full pass keeping what reaches the bound ---------------------------------------------------------------------------
    RL_HIP(hipMemsetAsync(cnt, 0, ((size_t)B + 1) * sizeof(uint32_t), s));  // list lengths + the overflow flag; the lists need no fill
    const CandArgs ca{top_s + (k - 1), k, c_s, c_i, cnt, flag, cap};
    idx->filt = {RL_FILTER_ROWS_FUSED, B, cap, cnt, flag};
    RL_TRY(launch_score_planes_pass(idx->planes.p, n, idx->dim, B, qs, nullptr, 0, idx->norm, idx->sumsq, mode, 1, nullptr, &ca, idx->n_cu, s,
                                    img_scale, half));
    // ---- (3) exact ranking of every list ------------------------------------------------------------------------------------------
    RL_TRY(launch_merge_topk(c_s, c_i, 1, B, cap, k, d_scores, d_rows, s, cnt));
    // ---- (4) guarded dense fallback -----------------------------------------------------------------------------------------------
    RL_TRY(launch_score_planes_pass(idx->planes.p, n, idx->dim, B, qs, sc, ld, idx->norm, idx->sumsq, mode, 1, flag, nullptr, idx->n_cu, s,
                                    img_scale, half));
    RL_TRY(launch_guarded_select(sc, B, n, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, d_scores, d_rows, flag, s));
    return RL_OK;
}

// The fused top-k of search_rows_fused with its two GEMM passes over the HI image at ONE fp16 MFMA product per multiply (q_hi . e_hi:
// a third of the matrix work and half the bytes of the pre-split image), made exact by the error band of search_rows_hi -- the default
// for big batches over an index that keeps a HI image since round 3 (BASELINE cfg 5: 7.48 -> 3.74 ms per 1000 queries on the same box,
// profiles/r03_a / r03_f; RAGLITE_NO_FUSED_HI=1, read per call, restores search_rows_fused: A/B):
//   (1) sample pass -> the k-th best APPROXIMATE similarity of a row subset, tau_s <= the k-th best approximate overall (A_k);
//   (2) every row of the exact top-k has an approximate similarity >= A_k - 2 m >= tau_s - 2 m (|approximate - exact| <= m, with
//       m from what the corpus' and the query's hi halves drop: row_threshold_kernel), so the candidate pass keeps the rows
//       that reach tau_s - 2 m: ~k * stride per query, in lists of (approximate similarity, row);
//   (3) each list sorted: its k-th entry IS A_k, and the entries >= A_k - 2 m -- a prefix, k + a few dozen rows -- are the only
//       rows that can be in the exact top-k (list_prefix_kernel);
//   (4) their exact similarities (row_dots_kernel, fp32) ranked by (score desc, row asc);
//   (5) any overflow / unusable threshold -> device flag -> the dense full-precision GEMM + selection, launched always, returning at
//       once when the flag is clear.
// Results: exact top-k of exactly computed fp32 similarities (integer data: bit-identical to the oracle; float data: the last
// bits differ from the dense path's split-arithmetic sums, as they do between any two of the paths).  cosine / dot, k <= 512,
// no row mask; RL_ERR_UNSUPPORTED otherwise.
int search_rows_fused_hi(rl_index* idx, const float* d_q, int32_t B, int32_t k, float* d_scores, int32_t* d_rows, int64_t ld, hipStream_t s) {
    if (!idx->opt.on(RL_OPT_FUSED_HI) || !idx->opt.on(RL_OPT_FUSED_TOPK)) return RL_ERR_UNSUPPORTED;  // (the second asks for the dense path: no fused top-k at all)
    const bool hi_only = true;  // (two products -- q_hi.e_hi + q_lo.e_hi, no |q_lo| term in the band -- measured 4.88 ms against 3.51)
    const int mode = scan_mode(idx->metric);
    if (B < rows_gemm_min(idx) || k > 512 || (mode != SCAN_COSINE && mode != SCAN_DOT)) return RL_ERR_UNSUPPORTED;
    if (!hi_image_valid(idx) || !image_valid(idx) || !idx->E) return RL_ERR_UNSUPPORTED;
    if (mode == SCAN_COSINE && !idx->norm) return RL_ERR_UNSUPPORTED;
    const int64_t n = idx->n_rows, T = (n + 255) / 256;
    const int32_t cap = MERGE_CAP, cap2 = 1024;
    int32_t stride = (int32_t)std::min<int64_t>(MERGE_CAP / (3 * (int64_t)k), T / 8);
    if (idx->opt.v[RL_OPT_FUSED_TOPK_STRIDE] > 0) stride = (int32_t)std::min<int64_t>(idx->opt.v[RL_OPT_FUSED_TOPK_STRIDE], T / 8);  // A/B
    if (stride < 2) return RL_ERR_UNSUPPORTED;
    const int64_t Tv = (T + stride - 1) / stride, ld_s = Tv * 256;
    if (ld_s < k) return RL_ERR_UNSUPPORTED;
    // ---- scratch ------------------------------------------------------------------------------------------------------------
    RL_TRY(idx->misc.reserve(score_planes_scratch_floats(B, idx->dim) * sizeof(float)));
    const size_t n_sample = (size_t)B * ld_s, n_top = (size_t)B * k, n_cand = (size_t)B * cap, n_c2 = (size_t)B * cap2;
    RL_TRY(idx->fused.reserve((n_sample + 2 * n_top + 2 * n_cand + 2 * n_c2 + 5 * (size_t)B + 16) * 4));
    float* S_s = idx->fused.as<float>();
    float* top_s = S_s + n_sample;
    int32_t* top_i = reinterpret_cast<int32_t*>(top_s + n_top);
    float* c_s = reinterpret_cast<float*>(top_i + n_top);
    int32_t* c_i = reinterpret_cast<int32_t*>(c_s + n_cand);
    int32_t* r_i = c_i + n_cand;                                   // [B x cap2] rows to re-score
    float* r_s = reinterpret_cast<float*>(r_i + n_c2);             // [B x cap2] their exact similarities
    float* thr = r_s + n_c2;                                       // [B]
    float* window = thr + B;                                       // [B]
    float* thr1 = window + B;                                      // [B] the first round's thresholds (two-round candidate pass)
    uint32_t* cnt = reinterpret_cast<uint32_t*>(thr1 + B);         // [B]
    uint32_t* cnt2 = cnt + B;                                      // [B]
    uint32_t* flag = cnt2 + B;
    float* qs = idx->misc.as<float>();
    const int32_t groups = (B + 31) / 32;
    const float* q_unscale = qs + (size_t)groups * 32 * idx->dim;  // (the layout of launch_score_planes_queries)
    const float* q_sumsq = q_unscale + B + groups;
    float* sc = idx->scores.as<float>();  // [B x ld], reserved by the caller: only the fallback touches it
    const void* hi = idx->hi_image.p;
    const float sscale = idx->split_scale;
    // ---- (1) sample pass + its exact top-k ------------------------------------------------------------------------------------
    RL_TRY(launch_score_planes_queries(d_q, B, idx->dim, qs, mode, s, flag));  // (also zeroes the flag)
    // (round 5: on the sixteen-group tile of maxsim_pp.hip, MODE 1 -- the eight-group kernel gives every workgroup ONE 256 x 256 tile and is
    // start-up bound there; RL_OPT_FUSED_PP_SAMPLE = 0 or shapes outside the tile: the eight-group kernel)
    int st_sample = RL_ERR_UNSUPPORTED;
    if (idx->opt.on(RL_OPT_FUSED_PP) && idx->opt.on(RL_OPT_FUSED_PP_SAMPLE) && hi_only && idx->dim % 32 == 0 && idx->dim >= 256)
        st_sample = launch_pp_rows_sample(hi, n, idx->dim, B, qs, idx->norm, mode, S_s, ld_s, stride, idx->n_cu, s, sscale);
    if (st_sample == RL_ERR_UNSUPPORTED)
        st_sample = launch_score_planes_pass(hi, n, idx->dim, B, qs, S_s, ld_s, idx->norm, idx->sumsq, mode, stride, nullptr, nullptr, idx->n_cu, s, sscale,
                                             true, hi_only);
    RL_TRY(st_sample);
    RL_TRY(launch_topk(S_s, B, ld_s, ld_s, k, idx->ws, top_s, top_i, s));
    // ---- (2) thresholds lowered by the error band; candidate pass ---------------------------------------------------------------
    RL_TRY(launch_row_threshold(top_s, B, k, d_q, idx->dim, mode, hi_only ? q_unscale : nullptr, idx->max_lo_ratio, idx->max_lo_norm, idx->max_row_norm,
                                thr, window, cnt, cnt2, flag, s, thr1, sum_eps(idx->dim)));  // (thr1: the first round's thresholds, kept for rl_time_kernel's replay)
    const CandArgs ca{thr, 1, c_s, c_i, cnt, flag, cap};
    idx->filt = {RL_FILTER_ROWS_FUSED_HI, B, cap, cnt, flag};
    // The candidate pass on the sixteen-group tile of maxsim_pp.hip (round 4: 128 rows x 512 queries per workgroup, every operand through
    // LDS-DMA rings, wave-private record logs; dim % 32 == 0, dim >= 256; RL_OPT_FUSED_PP = 0: the eight-group tile of maxsim_gemm.hip)
    // In TWO rounds: the threshold of the first comes from the row sample (every stride-th tile: ~k * stride rows per query reach it);
    // after 3/16 of the row tiles every list holds its query's candidates among those rows, and the k-th best of THEM -- a subset 5 x the
    // sample -- minus the band is a valid, much tighter threshold for the other 13/16 (the k-th best of any subset bounds the k-th best
    // overall from below): ~2.8 x fewer records to log, flush and rank (expected 2 700 f + 100 (1 - f) / f per query at k = 100, least at
    // f = 0.19).  Both rounds append to the same lists.
    int st_pp = RL_ERR_UNSUPPORTED;
    int64_t round1_tiles = 0;
    bool row_test = false;
    if (idx->opt.on(RL_OPT_FUSED_PP) && idx->dim % 32 == 0 && idx->dim >= 256) {
        int32_t log_cap = 0;
        const size_t work_bytes = pp_rows_scratch_bytes(n, B, idx->n_cu, (int32_t)std::min<int64_t>((int64_t)k * stride, cap), &log_cap);
        RL_TRY(idx->pp_work.reserve(work_bytes));
        const int64_t Tr = (n + 127) / 128;
        // cosines over rows whose norms span more than a factor of four: the candidate pass tests its hits row by row (maxsim_pp.hip: the
        // ROW-NORM variant) -- the block-wide bound would pass most of a block that holds a short row next to long ones
        row_test = mode == SCAN_COSINE && !(idx->min_row_norm * 4.0f >= idx->max_row_norm);
        round1_tiles = (Tr >= 64 && idx->opt.on(RL_OPT_FUSED_TWO_ROUNDS)) ? (3 * Tr) / 16 : 0;  // (small corpora: one round)
        if (round1_tiles > 0) {
            st_pp = launch_pp_rows_pass(hi, n, idx->dim, B, qs, idx->norm, mode, &ca, idx->pp_work.p, log_cap, idx->n_cu, s, sscale, 0, round1_tiles, false,
                                        row_test);
            if (st_pp == RL_OK) {
                if (idx->opt.on(RL_OPT_LIST_SELECT)) {  // (round 5: the k-th best of every list by a radix select, threshold raised in the same launch)
                    RL_TRY(launch_list_raise_threshold(c_s, c_i, B, cap, k, cnt, window, thr, s));
                } else {
                    RL_TRY(launch_merge_topk(c_s, c_i, 1, B, cap, k, top_s, top_i, s, cnt));  // (top_s / top_i: the sample's top-k is not needed any more)
                    RL_TRY(launch_raise_threshold(thr, top_s, B, k, window, s));
                }
                st_pp = launch_pp_rows_pass(hi, n, idx->dim, B, qs, idx->norm, mode, &ca, idx->pp_work.p, log_cap, idx->n_cu, s, sscale, round1_tiles,
                                            Tr - round1_tiles, true, row_test);
            }
        } else {
            st_pp = launch_pp_rows_pass(hi, n, idx->dim, B, qs, idx->norm, mode, &ca, idx->pp_work.p, log_cap, idx->n_cu, s, sscale, 0, -1, false, row_test);
        }
        if (st_pp != RL_OK && st_pp != RL_ERR_UNSUPPORTED) return st_pp;
    }
    if (st_pp != RL_OK)
        RL_TRY(launch_score_planes_pass(hi, n, idx->dim, B, qs, nullptr, 0, idx->norm, idx->sumsq, mode, 1, nullptr, &ca, idx->n_cu, s, sscale, true, hi_only));
    {
        auto& r = idx->replay;
        r.valid = true; r.pp = st_pp == RL_OK; r.B = B; r.mode = mode; r.qs = qs; r.ca = ca; r.cnt = cnt; r.thr1 = thr1; r.round1_tiles = round1_tiles; r.row_test = row_test;
        r.pools[0] = idx->misc.p; r.pools[1] = idx->fused.p; r.pools[2] = idx->pp_work.p;
        int32_t lc = 0;
        (void)pp_rows_scratch_bytes(n, B, idx->n_cu, (int32_t)std::min<int64_t>((int64_t)k * stride, cap), &lc);
        r.log_cap = lc;
    }
    // ---- (3) the rows within the band of each list's k-th entry; (4) their exact similarities, ranked -----------------------------
    if (idx->opt.on(RL_OPT_LIST_SELECT)) RL_TRY(launch_list_select(c_s, c_i, B, cap, k, cnt, window, cap2, r_i, cnt2, flag, s));
    else RL_TRY(launch_list_prefix(c_s, c_i, B, cap, k, cnt, window, cap2, r_i, cnt2, flag, s));
    RL_TRY(launch_row_dots(idx->E, idx->dim, d_q, B, r_i, cnt2, cap2, mode, idx->norm, q_sumsq, r_s, s));
    RL_TRY(launch_merge_topk(r_s, r_i, 1, B, cap2, k, d_scores, d_rows, s, cnt2));
    // ---- (5) guarded dense fallback (full precision, over the pre-split image) ------------------------------------------------------
    RL_TRY(launch_score_planes_pass(idx->planes.p, n, idx->dim, B, qs, sc, ld, idx->norm, idx->sumsq, mode, 1, flag, nullptr, idx->n_cu, s,
                                    image_scale(idx), false));
    // (its selection in ONE guarded launch, a block per query: the three launches of the selection, each a grid of 64 x B workgroups that return
    // at once behind the flag, were 36 us of every 1000-query batch)
    RL_TRY(launch_guarded_select(sc, B, n, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, d_scores, d_rows, flag, s));
    return RL_OK;
}

// Exact row top-k of up to 16 queries at HALF the bytes of a corpus pass (BASELINE cfg 2, `ORDER BY dist LIMIT k` of
// src/raglite/_search.py:69-79 for one query): the single-query search is HBM-bound on the 4 B per element of the fp32 corpus,
// and the hi halves of the fp16 split carry 11 of every element's 24 significand bits in 2 B.
//   (1) the f16 stream kernel over the HI plane -> approximate dots (query in full hi + lo precision), the metric, their exact
//       top-k;
//   (2) |approximate - exact| <= m for every row, rigorously: the dropped lo half is at most 2^-11 of its element (nearest), so
//       the dot product moves by <= 2^-10 |e| |q| (Cauchy-Schwarz) -- m = 2^-10 for a cosine, 2^-10 sqrt(dim) max|e| |q| for a
//       dot product, plus 2^-11 for the fp32 roundings of both passes.  A row can belong to the exact top-k only if its
//       approximate score reaches (k-th best approximate) - 2 m: one more pass over the 4 B per row of approximate scores
//       collects exactly those rows (k + a few dozen on embedding-like data; a list holds 1024);
//   (3) the candidates' fp32 rows are gathered and scored by the SAME kernels the full pass uses (a row's score does not
//       depend on where the row sits: tests/test_gpu_parity.py), ranked by (score desc, row asc): the same bits as the
//       full pass, ties included;
//   (4) list overflow (more than 1024 rows within 2 m of the k-th: near-duplicate-heavy corpora): the full-precision pass,
//       launched always, returning at once when the flag is clear -- nothing here synchronises with the host.
// cosine / dot, B <= 16 (measured at 1 M x 1024: 1.6x at B = 1, 1.3x at B = 16, slower at B = 32 where the gather and the
// re-scoring of 32 x 1024 candidate rows cost more than the half pass saves), k <= 512, with or without a row mask (masked rows rank -inf in the approximate pass: they are never
// candidates); RL_ERR_UNSUPPORTED otherwise.
int search_rows_hi(rl_index* idx, const float* d_q, int32_t nb, int32_t k, float* d_scores, int32_t* d_rows, int64_t ld, hipStream_t s,
                   const uint32_t* d_row_bits) {
    if (!idx->opt.on(RL_OPT_HI_SEARCH)) return RL_ERR_UNSUPPORTED;
    const int mode = scan_mode(idx->metric);
    const int64_t n = idx->n_rows;
    if (!hi_valid(idx) || nb > 16 || k > 512 || (mode != SCAN_COSINE && mode != SCAN_DOT && mode != SCAN_L2) || n < 65536) return RL_ERR_UNSUPPORTED;
    // l2 (round 6): 1 - |e - q|.  The approximate similarity comes from |e|^2 + |q|^2 - 2 e_hi.q, its bound lives on the SQUARED distance
    // (hi_filter.hip: l2_delta / lower_threshold), the candidates and the guarded full pass are scored by the scan that sums (e - q)^2 directly --
    // the full-precision route's own kernel for up to four queries, so the same bits; candidates from the pivot route, or -- under a row mask --
    // the collecting flow behind a pivot selection.  Needs the measured norms (an index whose HI plane was built with them) and |e|^2 per row.
    const bool l2 = mode == SCAN_L2;
    // (l2 similarities of a big corpus share their exponent and leading mantissa bits: the radix selection's threshold bin holds the whole corpus
    // and the ranked flow's candidate lists overflow -- measured: every query fell back -- so l2 takes this route only where the pivot does the
    // selecting: k <= 512 and >= 3 k group maxima; beyond, the full-precision route, whose own selection is the pivot's for l2)
    if (l2 && (nb > 4 || !idx->sumsq || !idx->opt.on(RL_OPT_HI_PIVOT) || !pivot_route_takes(n, k))) return RL_ERR_UNSUPPORTED;
    // WIDE index (dim > 1024, round 6): the stream kernels stop at 1024 (a wave keeps its slice of the queries in registers) -- the approximate
    // pass is the packed VALU scan over the HI plane (scan16.hip: up to four queries per pass; the fp32 scan of such an index takes ONE), the
    // candidates and the guarded full pass go through the fp32 scan.  Up to four queries: beyond, the passes over the plane cost what the
    // full-precision scan does.
    const bool wide = idx->dim > 1024;
    if (wide && (nb > 4 || (reinterpret_cast<uintptr_t>(d_q) & 15))) return RL_ERR_UNSUPPORTED;
    const int32_t dim = idx->dim, cap = 1024;  // candidates per query (expected: k + a few dozen)
    const int64_t nc = (int64_t)nb * cap, ldx = nc;
    // ---- scratch: approximate top-k scores (+ unused ids), thresholds, counters + flag, candidate rows, their norms, the
    // gathered rows, the exact score block and its diagonal -------------------------------------------------------------------------
    const size_t words = (size_t)nb * k * 2 + 32 + 32 + (size_t)nc * 2 + 4 + (size_t)nc * dim + (size_t)nb * ldx + (size_t)nc + 2 + 2 * pivot_scratch_words(nb);
    RL_TRY(idx->hibuf.reserve(words * 4));
    float* ts = idx->hibuf.as<float>();                        // [nb x k]
    int32_t* ti = reinterpret_cast<int32_t*>(ts + (size_t)nb * k);
    float* thr = reinterpret_cast<float*>(ti + (size_t)nb * k);  // [nb] (16 words)
    float* mb = thr + 16;                                        // [nb] (16 words) the error bound of each query
    uint32_t* cnt = reinterpret_cast<uint32_t*>(mb + 16);        // [nb] counters, then the flag at cnt[16]
    uint32_t* flag = cnt + 16;
    int32_t* ci = reinterpret_cast<int32_t*>(cnt + 32);        // [nb x cap]
    float* gn = reinterpret_cast<float*>(ci + nc);             // [nb x cap]
    // (16-byte aligned whatever the parity of nb * k: the kernels that score the gathered rows take 16-byte loads -- misaligned, the stream kernel
    // DECLINED after the approximate pass had run, every search with an odd nb * k paying for both routes, and the scan fell back to scalar loads,
    // whose other order of summation costs the last bit of parity with the full pass; found by scripts/soak_pivot.py with l2 in, round 6)
    float* G = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(gn + nc) + 15) & ~uintptr_t(15));  // [nb * cap x dim]
    float* xs = G + (size_t)nc * dim;                          // [nb x nb * cap]
    float* es = xs + (size_t)nb * ldx;                         // [nb x cap]
    uint64_t* bmax = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(es + nc) + 7) & ~uintptr_t(7));  // [nb x 2048] group maxima (pivot route)
    float* sc = idx->scores.as<float>();
    // ---- (1) approximate pass over the HI plane ---------------------------------------------------------------------------------------------
    int st = wide ? launch_scan_rows16(idx->hiplane.as<uint16_t>(), n, dim, d_q, nb, nullptr, SCAN_RAW_DOT, sc, ld, s)
                  : launch_maxsim_stream16(idx->hiplane.as<uint16_t>(), n, dim, d_q, nb, idx->row_to_chunk, idx->offsets, idx->n_chunks, 1, sc, ld,
                                           idx->n_cu, s);
    if (st != RL_OK) return st;
    // ---- (2) its exact top-k, and every row that could be in the exact top-k of the full-precision scores ----------------------------------
    // The bound: what the HI halves drop is known exactly per row -- max |e_lo| / |e| (cosine) and max |e_lo| (dot) are kept by the
    // index (refresh_hi_image) -- plus 2^-12 |e| |q| for the query's own 2^-22 split and twice the worst case of a 1024-term fp32
    // sum (6e-5).  Without those maxima (no HI image on this index): the a-priori 2^-10 of the truncation, plus 2^-11.
    const bool measured = idx->max_row_norm_rows == idx->n_rows && idx->max_row_norm_scale == idx->hi_scale && idx->max_row_norm > 0.f;
    bool gathered = false;
    float m_rel = 0x1p-10f + 0x1p-11f, e_bound = std::sqrt((float)dim) * idx->max_abs;
    if (measured) {
        if (mode == SCAN_COSINE) m_rel = idx->max_lo_ratio + sum_eps(dim);
        else { m_rel = 1.0f; e_bound = idx->max_lo_norm + sum_eps(idx->dim) * idx->max_row_norm; }
    }
    if (l2) {
        if (!measured) return RL_ERR_UNSUPPORTED;
        m_rel = sum_eps(dim);  // (l2_delta's arguments: the rounding term, the dot's bound per |q|, max |e|)
    }
    if (d_row_bits) {  // tombstones / metadata filter: masked rows rank -inf, so they reach neither the top-k nor the candidates
        RL_TRY(launch_transform(sc, nb, n, ld, idx->norm, idx->sumsq, d_q, dim, mode, s, 1.0f / idx->hi_scale));
        RL_TRY(launch_mask_scores(sc, nb, n, ld, d_row_bits, s));
        int st_sel = RL_ERR_UNSUPPORTED;
        if (l2) st_sel = launch_topk_pivot(sc, nb, n, ld, k, idx->ws, ts, ti, s);  // (crowded scores: see above)
        if (st_sel != RL_OK && st_sel != RL_ERR_UNSUPPORTED) return st_sel;
        if (st_sel == RL_ERR_UNSUPPORTED) RL_TRY(launch_topk(sc, nb, n, ld, k, idx->ws, ts, ti, s));
        RL_TRY(launch_approx_threshold(ts, nb, k, d_q, dim, mode, m_rel, e_bound, thr, cnt, flag, s, idx->max_row_norm));
        RL_TRY(launch_collect_above(sc, nb, n, ld, thr, mode == SCAN_COSINE ? idx->norm : nullptr, cap, ci, gn, cnt, flag, s));
    } else {
        // Round 4 (cfg 2: thirteen launches of 4.6 - 9.9 us behind a 0.31 ms pass; now eight): the transform + histogram launch also zeroes
        // the candidate counters and the flag (cnt[0 .. 32)) and leaves each query's bound m; the selection's filter keeps what lies less
        // than 2 m below the threshold bin, and its final kernel -- which knows the k-th best -- lists every row within 2 m of it: no
        // threshold kernel, no collecting pass over the scores.
        HiBound bound;
        bound.m_out = mb; bound.m_rel = m_rel; bound.e_norm_bound = e_bound; bound.e_max = idx->max_row_norm;
        // Round 6 (option hi_pivot): no approximate RANKING at all -- the candidates are re-scored and ranked exactly anyway, so any lower bound
        // of the k-th best approximate similarity will do for the threshold: the k-th largest of ~500 workgroup maxima (hi_filter.hip:
        // transform_bmax_kernel / pivot_collect_kernel), two launches instead of the selection's three (k <= 512, >= 3 k maxima)
        int st_pv = RL_ERR_UNSUPPORTED;
        if (idx->opt.on(RL_OPT_HI_PIVOT))
            st_pv = launch_pivot_route(sc, nb, n, ld, k, idx->norm, idx->sumsq, d_q, dim, mode, 1.0f / idx->hi_scale, bmax, cnt, 32, bound, thr, cap, ci, gn,
                                       cnt, flag, s, idx->E, G, &gathered);
        if (st_pv != RL_OK && st_pv != RL_ERR_UNSUPPORTED) return st_pv;
        if (st_pv == RL_ERR_UNSUPPORTED) {
            RL_TRY(launch_transform_hist(sc, nb, n, ld, idx->norm, idx->sumsq, d_q, dim, mode, idx->ws, s, 1.0f / idx->hi_scale, nullptr, cnt, 32, &bound));
            HiEmit emit;
            emit.m = mb; emit.cap = cap; emit.ids = ci; emit.norms = gn; emit.row_norm = mode == SCAN_COSINE ? idx->norm : nullptr;
            emit.cnt = cnt; emit.flag = flag; emit.thr = thr; emit.l2 = l2 ? 1 : 0;
            RL_TRY(launch_topk(sc, nb, n, ld, k, idx->ws, ts, ti, s, nullptr, true, &emit));
        }
    }
    idx->filt = {RL_FILTER_ROWS_HI, nb, cap, cnt, flag};
    // ---- (3) exact scores of the candidates, by the kernels of the full pass (slots past a list's length are neither gathered nor
    // ranked: the pass multiplies whatever their rows of G hold) -- and (4), the guarded full-precision pass over the corpus, as the
    // second grid row of the SAME launch: it returns at once unless a list overflowed / a bound was unusable ---------------------------------
    if (!gathered) RL_TRY(launch_gather_rows(idx->E, false, dim, n, ci, nc, G, s, cnt, cap));  // (the pivot route's collection gathers on its way)
    StreamSecondJob full;
    full.D = idx->E; full.n_rows = n; full.out = sc; full.ld = ld; full.run_if = flag;
    if (l2) {  // 1 - sqrt(sum (e - q)^2) of the gathered rows and, guarded, of the corpus: the scan of the full-precision route, same bits
        st = launch_scan_rows(G, nc, dim, d_q, nb, nullptr, SCAN_L2, xs, ldx, s);
        if (st == RL_OK) st = launch_scan_rows(idx->E, n, dim, d_q, nb, nullptr, SCAN_L2, sc, ld, s, flag);
    } else if (wide) {  // raw dots of the gathered rows, then the guarded pass over the corpus: the fp32 scan, a query per pass
        st = launch_scan_rows(G, nc, dim, d_q, nb, nullptr, SCAN_RAW_DOT, xs, ldx, s);
        if (st == RL_OK) st = launch_scan_rows(idx->E, n, dim, d_q, nb, nullptr, SCAN_RAW_DOT, sc, ld, s, flag);
    } else {
        st = launch_maxsim_stream_two(G, nc, dim, d_q, nb, idx->row_to_chunk, idx->offsets, idx->n_chunks, xs, ldx, idx->n_cu, s, idx->split_scale, full);
    }
    if (st != RL_OK) {  // (a late decline: the route did not answer -- whoever does records itself)
        idx->filt = {};
        return st;
    }
    if (nb > 1) RL_TRY(launch_diag_blocks(xs, ldx, cap, nc, es, s));  // (one query: the block IS its diagonal)
    MergeTransform tr;  // (the metric transform of the re-scored candidates happens on the way into the ranking: transform_kernel's statements)
    tr.row_norm = gn; tr.queries = d_q; tr.dim = dim; tr.mode = mode;
    RL_TRY(launch_merge_topk(nb > 1 ? es : xs, ci, 1, nb, cap, k, d_scores, d_rows, s, cnt, l2 ? nullptr : &tr));  // (l2: already similarities)
    // ---- (4b) ... and its selection -------------------------------------------------------------------------------------------------------
    if (d_row_bits) {  // (the mask needs no guard: applied to scores nobody reads it changes nothing)
        if (!l2) RL_TRY(launch_transform(sc, nb, n, ld, idx->norm, idx->sumsq, d_q, dim, mode, s, 1.0f, flag));
        RL_TRY(launch_mask_scores(sc, nb, n, ld, d_row_bits, s));
        RL_TRY(launch_topk(sc, nb, n, ld, k, idx->ws, d_scores, d_rows, s, flag));
    } else if (l2) {
        RL_TRY(launch_guarded_select(sc, nb, n, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, d_scores, d_rows, flag, s));
    } else {  // transform + exact top-k in ONE guarded launch (a block per query: slow, and run once in a blue moon)
        RL_TRY(launch_guarded_select(sc, nb, n, ld, k, idx->norm, idx->sumsq, d_q, dim, mode, 1.0f, d_scores, d_rows, flag, s));
    }
    return RL_OK;
}

// The exact top-k of idx->scores [nb x ld] (+ for l2 batches the exact re-scoring of the hits): the tail of every row search.
int select_from_scores(rl_index* idx, const float* d_qb, int32_t nb, int32_t k, float* o_s, int32_t* o_r, int64_t ld, bool hist_done, hipStream_t s) {
    const int64_t n = idx->n_rows;
    // l2 similarities (1 - |e - q|) of a big corpus crowd into one bin of the radix selection (its one-block slow path: 2 ms per query at 1 M
    // rows, three times the scan that produced the scores): the pivot route does not care how the scores are distributed (round 6)
    int st_pv = RL_ERR_UNSUPPORTED;
    if (!hist_done && idx->metric == RL_L2 && idx->opt.on(RL_OPT_HI_PIVOT))
        st_pv = launch_topk_pivot(idx->scores.as<float>(), nb, n, ld, k, idx->ws, o_s, o_r, s);
    if (st_pv != RL_OK && st_pv != RL_ERR_UNSUPPORTED) return st_pv;
    if (st_pv == RL_ERR_UNSUPPORTED) RL_TRY(launch_topk(idx->scores.as<float>(), nb, n, ld, k, idx->ws, o_s, o_r, s, nullptr, hist_done));
    if (idx->metric == RL_L2 && (nb > 4)) {
        // The batched paths rank by |e|^2 + |q|^2 - 2 e.q; re-score the k hits of every query with the exact
        // sum (e - q)^2 and re-sort them (near-duplicates would otherwise report a cancelled distance).
        const int64_t items = (int64_t)nb * k;
        RL_TRY(idx->misc.reserve((size_t)items * (sizeof(float) + 2 * sizeof(int32_t))));
        float* re = idx->misc.as<float>();
        int32_t* pos = reinterpret_cast<int32_t*>(re + items);
        int32_t* tmp_rows = pos + items;
        RL_TRY(launch_rescore_l2(idx->E16 ? (const void*)idx->E16 : (const void*)idx->E, idx->E16 != nullptr, idx->dim, d_qb, o_r, o_s, k, items,
                                 re, s));
        RL_HIP(hipMemcpyAsync(tmp_rows, o_r, (size_t)items * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        RL_TRY(launch_topk(re, nb, k, k, k, idx->ws, o_s, pos, s));
        RL_TRY(launch_permute_rows(tmp_rows, pos, k, items, o_r, s));
    }
    return RL_OK;
}

// rank_limit > 0: the order-first-then-filter branch (src/raglite/_search.py:120-141) -- only the rank_limit nearest LIVE
// rows of a query are eligible, and among those the rows d_row_bits lets through are ranked.
int search_rows_device(rl_index* idx, const float* d_q, int32_t B, int32_t k, float* d_scores, int32_t* d_rows,
                       hipStream_t s, const uint32_t* d_row_bits = nullptr, int64_t rank_limit = 0) {
    const int64_t n = idx->n_rows;
    const int64_t ld = (n + 3) & ~int64_t(3);
    if (n == 0) {  // empty index: every slot is padding (the reference returns ([], []), tests/test_search.py:76-85)
        RL_TRY(launch_fill_f32(d_scores, -std::numeric_limits<float>::infinity(), (int64_t)B * k, s));
        RL_HIP(hipMemsetAsync(d_rows, 0xff, (size_t)B * k * sizeof(int32_t), s));
        return RL_OK;
    }
    const int64_t per_query = std::max<int64_t>(ld * 4, 1);
    const int32_t batch = (int32_t)std::max<int64_t>(1, std::min<int64_t>(B, (int64_t)(SCORE_BATCH_BYTES / per_query)));
    RL_TRY(idx->scores.reserve((size_t)batch * ld * sizeof(float)));
    for (int32_t b0 = 0; b0 < B; b0 += batch) {
        const int32_t nb = std::min<int32_t>(batch, B - b0);
        const bool cut = rank_limit > 0 && rank_limit < n;
        // lazy images: the routes below test what the index HAS; ask for what this batch's route reads first
        if (!d_row_bits && !cut && nb >= rows_gemm_min(idx) && k <= 512 && idx->opt.on(RL_OPT_FUSED_TOPK))
            RL_TRY(demand_images(idx, IMG_PLANES | (idx->opt.on(RL_OPT_FUSED_HI) ? IMG_HI_IMAGE : 0u), s));
        if (!cut && nb <= ((idx->dim > 1024 || idx->metric == RL_L2) ? 4 : 16) && k <= 512 && idx->opt.on(RL_OPT_HI_SEARCH))
            RL_TRY(demand_images(idx, IMG_HI_PLANE, s));
        if (!d_row_bits && !cut) {  // big batches over an index with a HI image: fused top-k at one MFMA product per multiply
            const int st = search_rows_fused_hi(idx, d_q + (int64_t)b0 * idx->dim, nb, k, d_scores + (int64_t)b0 * k, d_rows + (int64_t)b0 * k, ld, s);
            if (st == RL_OK) continue;
            if (st != RL_ERR_UNSUPPORTED) return st;
        }
        if (!d_row_bits && !cut) {  // big batches over the pre-split image: no score matrix at all
            const int st = search_rows_fused(idx, d_q + (int64_t)b0 * idx->dim, nb, k, d_scores + (int64_t)b0 * k, d_rows + (int64_t)b0 * k, ld, s);
            if (st == RL_OK) continue;
            if (st != RL_ERR_UNSUPPORTED) return st;
        }
        if (!cut && nb <= 16) {  // a few queries over a big fp32 corpus: half the bytes through the HI plane
            const int st = search_rows_hi(idx, d_q + (int64_t)b0 * idx->dim, nb, k, d_scores + (int64_t)b0 * k, d_rows + (int64_t)b0 * k, ld, s,
                                          d_row_bits);
            if (st == RL_OK) continue;
            if (st != RL_ERR_UNSUPPORTED) return st;
        }
        bool hist_done = false;
        RL_TRY(score_rows(idx, d_q + (int64_t)b0 * idx->dim, nb, ld, s, (d_row_bits || cut) ? nullptr : &hist_done));
        if (cut) {  // tombstoned rows are not in the reference's table at all: out before the cut, then cut + filter
            if (idx->live_row_bits) RL_TRY(launch_mask_scores(idx->scores.as<float>(), nb, n, ld, idx->live_row_bits, s));
            RL_TRY(idx->rankbuf.reserve(rank_cut_scratch_bytes(nb, n)));
            RL_TRY(launch_rank_cut(idx->scores.as<float>(), nb, n, ld, rank_limit, d_row_bits, idx->rankbuf.p, s));
        } else if (d_row_bits) {
            RL_TRY(launch_mask_scores(idx->scores.as<float>(), nb, n, ld, d_row_bits, s));
        }
        RL_TRY(select_from_scores(idx, d_q + (int64_t)b0 * idx->dim, nb, k, d_scores + (int64_t)b0 * k, d_rows + (int64_t)b0 * k, ld, hist_done, s));
    }
    if (d_row_bits || (rank_limit > 0 && rank_limit < n)) RL_TRY(launch_fix_masked(d_scores, d_rows, (int64_t)B * k, s));  // masked rows are "no hit"
    return RL_OK;
}

int check_search_args(const rl_index* idx, const float* q, int32_t B, int32_t k, const char* who) {
    if (!idx) return fail(RL_ERR_INVALID, std::string(who) + ": null index");
    if (B < 0 || k < 1) return fail(RL_ERR_INVALID, std::string(who) + ": n_queries must be >= 0 and k >= 1");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, std::string(who) + ": k must be <= 2048");
    if (B > 0 && !q) return fail(RL_ERR_INVALID, std::string(who) + ": null queries");
    return RL_OK;
}

}  // namespace

int rl_search_rows_ranked(rl_index* idx, const float* queries, int32_t B, int32_t k, const uint32_t* chunk_filter,
                          int64_t rank_limit, float* out_scores, int32_t* out_rows, int mem, void* stream) {
    RL_TRY(check_search_args(idx, queries, B, k, "rl_search_rows"));
    if (rank_limit < 0) return fail(RL_ERR_INVALID, "rl_search_rows: rank_limit must be >= 0 (0 = no cut)");
    if (B == 0) return RL_OK;
    if (!out_scores || !out_rows) return fail(RL_ERR_INVALID, "rl_search_rows: null output");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_s, t_r, t_f;
    const float* d_q; float* d_s; int32_t* d_r;
    const uint32_t* d_f = nullptr;
    const uint32_t* d_bits = nullptr;
    if (chunk_filter) RL_TRY(stage_in(chunk_filter, (size_t)((idx->n_chunks + 31) / 32), mem, s, t_f, &d_f));
    RL_TRY(effective_row_mask(idx, d_f, s, &d_bits));
    RL_TRY(stage_in(queries, (size_t)B * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_scores, (size_t)B * k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_rows, (size_t)B * k, mem, t_r, &d_r));
    RL_TRY(search_rows_device(idx, d_q, B, k, d_s, d_r, s, d_bits, rank_limit));
    RL_TRY(stage_out_end(out_scores, (size_t)B * k, mem, s, t_s));
    RL_TRY(stage_out_end(out_rows, (size_t)B * k, mem, s, t_r));
    return finish(mem, s);
}

int rl_search_rows_filtered(rl_index* idx, const float* queries, int32_t B, int32_t k, const uint32_t* chunk_filter,
                            float* out_scores, int32_t* out_rows, int mem, void* stream) {
    return rl_search_rows_ranked(idx, queries, B, k, chunk_filter, 0, out_scores, out_rows, mem, stream);
}

int rl_search_rows(rl_index* idx, const float* queries, int32_t B, int32_t k, float* out_scores, int32_t* out_rows,
                   int mem, void* stream) {
    return rl_search_rows_filtered(idx, queries, B, k, nullptr, out_scores, out_rows, mem, stream);
}

// ---- the order-first cut over a SHARDED corpus, in stages (include/raglite_hip.h) ---------------------------------------------------------
int rl_rank_cut_begin(rl_index* idx, const float* queries, int32_t B, int mem, void* stream) {
    RL_TRY(check_search_args(idx, queries, B, 1, "rl_rank_cut_begin"));
    if (B < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_begin: need at least one query");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    idx->rank_B = 0;
    const int64_t n = idx->n_rows, ld = std::max<int64_t>((n + 3) & ~int64_t(3), 4);
    if ((int64_t)B * ld * 4 > (int64_t)SCORE_BATCH_BYTES) return fail(RL_ERR_UNSUPPORTED, "rl_rank_cut_begin: too many queries for one score batch (split the batch)");
    DevBuf t_q;
    const float* d_q;
    RL_TRY(stage_in(queries, (size_t)B * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(idx->rank_q.reserve((size_t)B * idx->dim * sizeof(float)));
    RL_HIP(hipMemcpyAsync(idx->rank_q.p, d_q, (size_t)B * idx->dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    RL_TRY(idx->scores.reserve((size_t)B * ld * sizeof(float)));
    RL_TRY(idx->rankbuf.reserve(rank_stage_scratch_bytes(B, std::max<int64_t>(n, 1))));
    if (n > 0) {
        RL_TRY(score_rows(idx, idx->rank_q.as<float>(), B, ld, s, nullptr));
        if (idx->live_row_bits) RL_TRY(launch_mask_scores(idx->scores.as<float>(), B, n, ld, idx->live_row_bits, s));  // tombstones are not in the table
    }
    idx->rank_B = B;
    return finish(mem, s);
}

namespace {
uint32_t* rank_level_buf(rl_index* idx) { return reinterpret_cast<uint32_t*>(idx->rankbuf.as<char>() + rank_cut_scratch_bytes(idx->rank_B, std::max<int64_t>(idx->n_rows, 1))); }
}

int rl_rank_cut_level(rl_index* idx, int level, int64_t rank_limit, uint32_t* out_hist, int mem, void* stream) {
    if (!idx || !out_hist) return fail(RL_ERR_INVALID, "rl_rank_cut_level: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    if (idx->rank_B < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_level: no rl_rank_cut_begin in progress on this index");
    if (level < 0 || level > 2 || rank_limit < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_level: level must be 0..2 and rank_limit >= 1");
    const int64_t n = idx->n_rows, ld = std::max<int64_t>((n + 3) & ~int64_t(3), 4);
    DevBuf t_o;
    uint32_t* d_o;
    const size_t words = (size_t)idx->rank_B * HIST_BINS;
    RL_TRY(stage_out_begin(out_hist, words, mem, t_o, &d_o));
    uint32_t* lvl = mem == RL_MEM_DEVICE ? d_o : rank_level_buf(idx);
    RL_TRY(launch_rank_stage_level(idx->scores.as<float>(), idx->rank_B, n, ld, rank_limit, level, idx->rankbuf.p, lvl, s));
    if (lvl != d_o) RL_HIP(hipMemcpyAsync(d_o, lvl, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    RL_TRY(stage_out_end(out_hist, words, mem, s, t_o));
    return finish(mem, s);
}

int rl_rank_cut_level_done(rl_index* idx, int level, const uint32_t* hist_sum, int mem, void* stream) {
    if (!idx || !hist_sum) return fail(RL_ERR_INVALID, "rl_rank_cut_level_done: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    if (idx->rank_B < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_level_done: no rl_rank_cut_begin in progress on this index");
    if (level < 0 || level > 2) return fail(RL_ERR_INVALID, "rl_rank_cut_level_done: level must be 0..2");
    DevBuf t_i;
    const uint32_t* d_i;
    RL_TRY(stage_in(hist_sum, (size_t)idx->rank_B * HIST_BINS, mem, s, t_i, &d_i));
    RL_TRY(launch_rank_stage_set_level(idx->rank_B, level, idx->rankbuf.p, d_i, s));
    return finish(mem, s);
}

int rl_rank_cut_ties(rl_index* idx, int64_t rank_limit, uint32_t* out_ties, int mem, void* stream) {
    if (!idx || !out_ties) return fail(RL_ERR_INVALID, "rl_rank_cut_ties: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    if (idx->rank_B < 1 || rank_limit < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_ties: no rl_rank_cut_begin in progress on this index");
    const int64_t n = idx->n_rows, ld = std::max<int64_t>((n + 3) & ~int64_t(3), 4);
    DevBuf t_o;
    uint32_t* d_o;
    RL_TRY(stage_out_begin(out_ties, (size_t)idx->rank_B, mem, t_o, &d_o));
    RL_TRY(launch_rank_stage_ties(idx->scores.as<float>(), idx->rank_B, n, ld, rank_limit, idx->rankbuf.p, d_o, s));
    RL_TRY(stage_out_end(out_ties, (size_t)idx->rank_B, mem, s, t_o));
    return finish(mem, s);
}

int rl_rank_cut_finish(rl_index* idx, int64_t rank_limit, const uint32_t* ties_before, const uint32_t* chunk_filter, int32_t k,
                       float* out_scores, int32_t* out_rows, int mem, void* stream) {
    if (!idx || !ties_before || !out_scores || !out_rows) return fail(RL_ERR_INVALID, "rl_rank_cut_finish: null argument");
    if (k < 1 || k > K_MAX || rank_limit < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_finish: k must be in [1, 2048] and rank_limit >= 1");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    const int32_t B = idx->rank_B;
    if (B < 1) return fail(RL_ERR_INVALID, "rl_rank_cut_finish: no rl_rank_cut_begin in progress on this index");
    idx->rank_B = 0;
    const int64_t n = idx->n_rows, ld = std::max<int64_t>((n + 3) & ~int64_t(3), 4);
    DevBuf t_b, t_f, t_s, t_r;
    const uint32_t* d_b; const uint32_t* d_f = nullptr; const uint32_t* d_bits = nullptr;
    float* d_s; int32_t* d_r;
    RL_TRY(stage_in(ties_before, (size_t)B, mem, s, t_b, &d_b));
    if (chunk_filter) RL_TRY(stage_in(chunk_filter, (size_t)((idx->n_chunks + 31) / 32), mem, s, t_f, &d_f));
    RL_TRY(effective_row_mask(idx, d_f, s, &d_bits));
    RL_TRY(stage_out_begin(out_scores, (size_t)B * k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_rows, (size_t)B * k, mem, t_r, &d_r));
    if (n == 0) {
        RL_TRY(launch_fill_f32(d_s, -std::numeric_limits<float>::infinity(), (int64_t)B * k, s));
        RL_HIP(hipMemsetAsync(d_r, 0xff, (size_t)B * k * sizeof(int32_t), s));
    } else {
        RL_TRY(launch_rank_stage_apply(idx->scores.as<float>(), B, n, ld, rank_limit, d_bits, idx->rankbuf.p, d_b, s));
        RL_TRY(select_from_scores(idx, idx->rank_q.as<float>(), B, k, d_s, d_r, ld, false, s));
        RL_TRY(launch_fix_masked(d_s, d_r, (int64_t)B * k, s));  // rows outside the cut / the filter are "no hit"
    }
    RL_TRY(stage_out_end(out_scores, (size_t)B * k, mem, s, t_s));
    RL_TRY(stage_out_end(out_rows, (size_t)B * k, mem, s, t_r));
    return finish(mem, s);
}

// ---- a6 + a7 + a8 --------------------------------------------------------------------------------------
int rl_search_chunks_ranked(rl_index* idx, const float* queries, int32_t B, int32_t num_hits, int32_t k,
                            const uint32_t* chunk_filter, int64_t rank_limit, float* out_scores, int32_t* out_chunks,
                            int32_t* out_counts, int mem, void* stream) {
    RL_TRY(check_search_args(idx, queries, B, k, "rl_search_chunks"));
    if (rank_limit < 0) return fail(RL_ERR_INVALID, "rl_search_chunks: rank_limit must be >= 0 (0 = no cut)");
    if (num_hits < 1 || num_hits > K_MAX) return fail(RL_ERR_INVALID, "rl_search_chunks: num_hits must be in [1, 2048]");
    if (B == 0) return RL_OK;
    if (!out_scores || !out_chunks || !out_counts) return fail(RL_ERR_INVALID, "rl_search_chunks: null output");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_s, t_c, t_n, t_f;
    const float* d_q; float* d_s; int32_t* d_c; int32_t* d_n;
    const uint32_t* d_f = nullptr;
    const uint32_t* d_bits = nullptr;
    if (chunk_filter) RL_TRY(stage_in(chunk_filter, (size_t)((idx->n_chunks + 31) / 32), mem, s, t_f, &d_f));
    RL_TRY(effective_row_mask(idx, d_f, s, &d_bits));
    RL_TRY(stage_in(queries, (size_t)B * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_scores, (size_t)B * k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_chunks, (size_t)B * k, mem, t_c, &d_c));
    RL_TRY(stage_out_begin(out_counts, (size_t)B, mem, t_n, &d_n));
    RL_TRY(idx->hits.reserve((size_t)B * num_hits * 8));
    float* h_s = idx->hits.as<float>();
    int32_t* h_r = reinterpret_cast<int32_t*>(h_s + (size_t)B * num_hits);
    RL_TRY(search_rows_device(idx, d_q, B, num_hits, h_s, h_r, s, d_bits, rank_limit));
    RL_TRY(launch_group_chunk_max(h_s, h_r, B, num_hits, idx->offsets, idx->n_chunks, k, d_s, d_c, d_n, s));
    RL_TRY(stage_out_end(out_scores, (size_t)B * k, mem, s, t_s));
    RL_TRY(stage_out_end(out_chunks, (size_t)B * k, mem, s, t_c));
    RL_TRY(stage_out_end(out_counts, (size_t)B, mem, s, t_n));
    return finish(mem, s);
}

int rl_search_chunks_filtered(rl_index* idx, const float* queries, int32_t B, int32_t num_hits, int32_t k,
                              const uint32_t* chunk_filter, float* out_scores, int32_t* out_chunks, int32_t* out_counts,
                              int mem, void* stream) {
    return rl_search_chunks_ranked(idx, queries, B, num_hits, k, chunk_filter, 0, out_scores, out_chunks, out_counts, mem, stream);
}

int rl_search_chunks(rl_index* idx, const float* queries, int32_t B, int32_t num_hits, int32_t k, float* out_scores,
                     int32_t* out_chunks, int32_t* out_counts, int mem, void* stream) {
    return rl_search_chunks_filtered(idx, queries, B, num_hits, k, nullptr, out_scores, out_chunks, out_counts, mem,
                                     stream);
}

// ---- a9 --------------------------------------------------------------------------------------------------
namespace {

// Chunk-level masking of MaxSim scores: the caller's filter, then the tombstones.
int mask_chunk_scores(rl_index* idx, float* d_scores, int32_t nb, int64_t ld, const uint32_t* d_chunk_filter,
                      hipStream_t s) {
    if (d_chunk_filter) RL_TRY(launch_mask_scores(d_scores, nb, idx->n_chunks, ld, d_chunk_filter, s));
    if (idx->live_chunk_bits) RL_TRY(launch_mask_scores(d_scores, nb, idx->n_chunks, ld, idx->live_chunk_bits, s));
    return RL_OK;
}

// Batched MaxSim in fp16-split arithmetic scores two queries per corpus pass (maxsim_stream2_kernel).  pairs_prepare
// splits `n_queries` queries into fp16 (hi, lo) MFMA fragments once (idx->qsplit); pairs_pass scores queries `first` and
// `first + 1` of them.  RL_ERR_UNSUPPORTED when the shape or the index' arithmetic does not allow it -- the caller then
// makes one pass per query.
int pairs_prepare(rl_index* idx, const float* d_q, int32_t nq, int64_t q_stride, int32_t n_queries, hipStream_t s) {
    if (!idx->opt.on(RL_OPT_QUERY_PAIRS) || idx->n_rows == 0 || idx->n_chunks == 0) return RL_ERR_UNSUPPORTED;
    if (!idx->E16 && !(idx->E && idx->split_scale > 0.f)) return RL_ERR_UNSUPPORTED;  // fp16-stored, or fp32 in split arithmetic
    if (nq <= 16 || nq > 32 || n_queries < 2) return RL_ERR_UNSUPPORTED;
    const int32_t d = idx->dim;
    if (d != 128 && d != 256 && d != 384 && d != 512 && d != 768 && d != 1024) return RL_ERR_UNSUPPORTED;
    RL_TRY(idx->qsplit.reserve(query_split_bytes(d, n_queries)));
    return launch_query_split(d_q, d, nq, q_stride, n_queries, idx->qsplit.as<char>(), idx->E16 != nullptr, s);
}

int pairs_pass(rl_index* idx, int32_t nq, int32_t n_queries, int32_t first, float* d_out, int64_t out_stride, hipStream_t s) {
    if (idx->has_empty_chunk) {
        RL_TRY(launch_fill_f32(d_out, -std::numeric_limits<float>::infinity(), idx->n_chunks, s));
        RL_TRY(launch_fill_f32(d_out + out_stride, -std::numeric_limits<float>::infinity(), idx->n_chunks, s));
    }
    return launch_maxsim_stream2(idx->E16 ? (const void*)idx->E16 : (const void*)idx->E, idx->E16 != nullptr, idx->n_rows, idx->dim,
                                 idx->qsplit.as<char>(), n_queries, first, nq, idx->row_to_chunk, idx->offsets, idx->n_chunks, d_out,
                                 out_stride, idx->n_cu, s, idx->split_scale);
}

// Batched MaxSim over the pre-split corpus image scores EIGHT queries per corpus pass (maxsim_gemm.hip).  gemm_prepare lays
// the queries' fp16 (hi, lo) MFMA fragments out once per batch (idx->qplanes); gemm_pass scores queries first .. first +
// n_q - 1 of them.  RL_ERR_UNSUPPORTED when the index has no image, has an empty chunk (the kernel finds a chunk by
// counting chunk ends) or the shape is outside the kernel -- the caller then uses the streaming kernels.
constexpr int32_t GEMM_PASS_QUERIES = 8, GEMM_PASS_MIN_QUERIES = 3;
// (a WIDE index -- dim > 1024 -- has no streaming MaxSim kernel to fall back on but the VALU backstop: even ONE query goes through the pass)
int32_t gemm_min_queries(const rl_index* idx) { return idx->dim > 1024 ? 1 : GEMM_PASS_MIN_QUERIES; }
// An index WITHOUT the pre-split image (RL_OPT_KEEP_IMAGE = 0: rows + HI image, 1.5 x the corpus) still runs the bound-filtered batch: the
// approximate pass reads the HI image, the exact re-scoring the rows, and the guarded full-precision fallback the rows through the streaming
// kernels (one launch, grid row = query) -- which is what decides the shapes this holds for.
bool slim_batch_ok(const rl_index* idx, const float* d_q) {
    const int32_t d = idx->dim;
    // (the guarded fallback without the pre-split image: the streaming kernels' dims, or -- wide indexes -- the exact kernel over every chunk)
    return !image_valid(idx) && !idx->E16 && idx->E && hi_image_valid(idx) &&
           (d == 256 || d == 384 || d == 512 || d == 768 || d == 1024 || (d > 1024 && hi_dim_ok(d))) &&
           !(reinterpret_cast<uintptr_t>(d_q) & 15) && !(reinterpret_cast<uintptr_t>(idx->E) & 15);
}
// zero_words: sixteen words the query-image kernel zeroes on its way (the batch's flag block: no memset launch)
int gemm_prepare(rl_index* idx, const float* d_q, int32_t nq, int64_t q_stride, int32_t n_queries, hipStream_t s, bool want_planes = false,
                 uint32_t* zero_words = nullptr) {
    if (!idx->opt.on(RL_OPT_GEMM_PASS)) return RL_ERR_UNSUPPORTED;
    if (idx->has_empty_chunk || idx->n_chunks == 0 || nq < 1 || nq > 32 || n_queries < gemm_min_queries(idx)) return RL_ERR_UNSUPPORTED;
    // lazy images: the image the approximate pass multiplies; the pre-split image only where the batch cannot run on rows + HI image
    // (options that route it through the eight-query kernel, shapes outside the slim batch, or the caller says so: k > 512, timing hooks)
    RL_TRY(demand_images(idx, approx_image_bit(idx), s));
    const bool hi_route = idx->opt.on(RL_OPT_HI_MAXSIM) && idx->opt.v[RL_OPT_HI_PRODUCTS] == 1 && idx->dim >= 256 && idx->opt.on(RL_OPT_PP_PASS);
    const bool fell_back = idx->h_fell_back && *static_cast<volatile uint32_t*>(idx->h_fell_back) != 0u;  // (the previous batch's flag, if it has arrived)
    if (want_planes || !hi_route || fell_back || (!image_valid(idx) && !slim_batch_ok(idx, d_q))) RL_TRY(demand_images(idx, IMG_PLANES, s));
    if (!(image_valid(idx) || slim_batch_ok(idx, d_q))) return RL_ERR_UNSUPPORTED;
    RL_TRY(idx->qplanes.reserve(query_planes_bytes(idx->dim, n_queries)));
    return launch_query_planes(d_q, idx->dim, nq, q_stride, n_queries, idx->qplanes.p, s, zero_words, zero_words ? 16 : 0);
}
int gemm_pass(rl_index* idx, int32_t nq, int32_t n_queries, int32_t first, int32_t n_q, float* d_out, int64_t out_stride, hipStream_t s) {
    return launch_maxsim_gemm(idx->planes.p, idx->n_rows, idx->dim, idx->qplanes.p, n_queries, first, n_q, nq, idx->row_to_chunk,
                              idx->offsets, idx->ends.as<uint32_t>(), d_out, out_stride, idx->n_cu, s, image_scale(idx), idx->E16 != nullptr);
}

int maxsim_scores_device(rl_index* idx, const float* d_q, int32_t nq, float* d_out, hipStream_t s) {
    if (idx->n_chunks == 0) return RL_OK;
    int st = RL_ERR_UNSUPPORTED;
    if (idx->has_empty_chunk || idx->n_rows == 0)
        RL_TRY(launch_fill_f32(d_out, -std::numeric_limits<float>::infinity(), idx->n_chunks, s));
    if (idx->n_rows == 0) return RL_OK;
    if (idx->E16 && idx->dim > 1024) {  // a wide fp16-stored index: the exact re-scoring kernel over every chunk (empty chunks: -inf)
        if (nq > 32) return fail(RL_ERR_UNSUPPORTED, "MaxSim over an fp16-stored index wider than 1024 takes up to 32 query vectors");
        st = launch_maxsim_pairs_all_wide(reinterpret_cast<const float*>(idx->E16), idx->dim, d_q, nq, (int64_t)nq * idx->dim, idx->offsets, idx->n_chunks, 1,
                                          d_out, idx->n_chunks, s, nullptr, true);
        if (st == RL_ERR_UNSUPPORTED) return fail(st, "MaxSim: query vectors must be 16-byte aligned on an fp16-stored index wider than 1024");
        return st;
    }
    // More than 32 query vectors: passes of 32, each adding its chunk scores to the previous ones (fixed pass order).
    for (int32_t v0 = 0; v0 < nq; v0 += 32) {
        const int32_t nv = std::min<int32_t>(32, nq - v0);
        const float* qv = d_q + (int64_t)v0 * idx->dim;
        const int64_t accumulate = v0 > 0 ? 1 : 0;
        st = idx->E16 ? launch_maxsim_stream16(idx->E16, idx->n_rows, idx->dim, qv, nv, idx->row_to_chunk, idx->offsets,
                                               idx->n_chunks, 0, d_out, accumulate, idx->n_cu, s)
                      : launch_maxsim_stream(idx->E, idx->n_rows, idx->dim, qv, nv, idx->row_to_chunk, idx->offsets,
                                             idx->n_chunks, 0, d_out, accumulate, idx->n_cu, s, idx->split_scale);
        if (st != RL_OK) break;
    }
    if (idx->E16 && st == RL_ERR_UNSUPPORTED) return fail(st, "MaxSim: unsupported shape for an fp16-stored index");
    if (st == RL_ERR_UNSUPPORTED)
        st = launch_maxsim_generic(idx->E, idx->dim, d_q, nq, (int64_t)nq * idx->dim, idx->offsets, nullptr,
                                   idx->n_chunks, 1, d_out, s);
    return st;
}

int maxsim_few_hi_plane(rl_index* idx, const float* d_q, int32_t nq, int32_t n, int32_t k, float* sc, int64_t ld, float* d_s, int32_t* d_c,
                        hipStream_t s, const uint32_t* d_filter = nullptr);  // (defined behind the batch pipeline it shares its stages with)
int maxsim_topk_batch_device(rl_index* idx, const float* d_q, bool q16, int32_t n_queries, int32_t nq, int32_t k, float* sc, int64_t ld, float* d_s,
                             int32_t* d_c, hipStream_t s, const uint32_t* d_filter = nullptr);
}  // namespace

int rl_maxsim_scores(rl_index* idx, const float* query_vecs, int32_t nq, float* out_scores, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_scores: null index");
    if (nq < 1 || !query_vecs) return fail(RL_ERR_INVALID, "rl_maxsim_scores: need at least one query vector");
    if (!out_scores) return fail(RL_ERR_INVALID, "rl_maxsim_scores: null output");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_o;
    const float* d_q; float* d_o;
    RL_TRY(stage_in(query_vecs, (size_t)nq * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_scores, (size_t)idx->n_chunks, mem, t_o, &d_o));
    RL_TRY(maxsim_scores_device(idx, d_q, nq, d_o, s));
    RL_TRY(mask_chunk_scores(idx, d_o, 1, idx->n_chunks, nullptr, s));  // deleted chunks score -inf
    RL_TRY(stage_out_end(out_scores, (size_t)idx->n_chunks, mem, s, t_o));
    return finish(mem, s);
}

int rl_maxsim_topk_filtered(rl_index* idx, const float* query_vecs, int32_t nq, int32_t k, const uint32_t* chunk_filter,
                            float* out_scores, int32_t* out_chunks, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_topk: null index");
    if (nq < 1 || !query_vecs) return fail(RL_ERR_INVALID, "rl_maxsim_topk: need at least one query vector");
    if (k < 1) return fail(RL_ERR_INVALID, "rl_maxsim_topk: k must be >= 1");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_topk: k must be <= 2048");
    if (!out_scores || !out_chunks) return fail(RL_ERR_INVALID, "rl_maxsim_topk: null output");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_s, t_c, t_f;
    const float* d_q; float* d_s; int32_t* d_c;
    const uint32_t* d_f = nullptr;
    if (chunk_filter) RL_TRY(stage_in(chunk_filter, (size_t)((idx->n_chunks + 31) / 32), mem, s, t_f, &d_f));
    RL_TRY(stage_in(query_vecs, (size_t)nq * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_scores, (size_t)k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_chunks, (size_t)k, mem, t_c, &d_c));
    RL_TRY(idx->scores.reserve(std::max<size_t>((size_t)idx->n_chunks * sizeof(float), 16)));
    if (idx->dim > 1024) {  // a WIDE index: the batch's routes (the bound-filtered pipeline where the index keeps a HI image), batch of one
        const int64_t ld1 = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
        RL_TRY(idx->scores.reserve((size_t)ld1 * sizeof(float)));
        RL_TRY(maxsim_topk_batch_device(idx, d_q, false, 1, nq, k, idx->scores.as<float>(), ld1, d_s, d_c, s, d_f));
        RL_TRY(stage_out_end(out_scores, (size_t)k, mem, s, t_s));
        RL_TRY(stage_out_end(out_chunks, (size_t)k, mem, s, t_c));
        return finish(mem, s);
    }
    {   // one user query at a time: the half-width route over the HI plane where the index has (or may build) one -- with a metadata filter too
        // (round 6: the filtered-out chunks rank -inf in the approximate scores, like tombstones)
        const int64_t ld1 = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
        RL_TRY(idx->scores.reserve((size_t)ld1 * sizeof(float)));
        const int st = maxsim_few_hi_plane(idx, d_q, nq, 1, k, idx->scores.as<float>(), ld1, d_s, d_c, s, d_f);
        if (st == RL_OK) {
            if (idx->live_chunk_bits || d_f) RL_TRY(launch_fix_masked(d_s, d_c, k, s));
            RL_TRY(stage_out_end(out_scores, (size_t)k, mem, s, t_s));
            RL_TRY(stage_out_end(out_chunks, (size_t)k, mem, s, t_c));
            return finish(mem, s);
        }
        if (st != RL_ERR_UNSUPPORTED) return st;
    }
    RL_TRY(maxsim_scores_device(idx, d_q, nq, idx->scores.as<float>(), s));
    const bool masked = d_f || idx->live_chunk_bits;
    RL_TRY(mask_chunk_scores(idx, idx->scores.as<float>(), 1, idx->n_chunks, d_f, s));
    RL_TRY(launch_topk(idx->scores.as<float>(), 1, idx->n_chunks, idx->n_chunks, k, idx->ws, d_s, d_c, s));
    if (masked) RL_TRY(launch_fix_masked(d_s, d_c, k, s));
    RL_TRY(stage_out_end(out_scores, (size_t)k, mem, s, t_s));
    RL_TRY(stage_out_end(out_chunks, (size_t)k, mem, s, t_c));
    return finish(mem, s);
}

int rl_maxsim_topk(rl_index* idx, const float* query_vecs, int32_t nq, int32_t k, float* out_scores,
                   int32_t* out_chunks, int mem, void* stream) {
    return rl_maxsim_topk_filtered(idx, query_vecs, nq, k, nullptr, out_scores, out_chunks, mem, stream);
}

namespace {
// The bound-filtered MaxSim batch in two halves, so that a SHARDED corpus can put one exchange between them
// (rl_maxsim_batch_begin / _finish): (a) approximate passes + the local top-k of the approximate scores, (b) given thr[] / cnt[] = 0:
// candidate collection, exact re-scoring, ranking, and the guarded full-precision fallback.
struct HiBatch {
    float* ts; int32_t* ti; float* thr; uint32_t* cnt; uint32_t* flag; int32_t* ci; float* es;
    float* m; float* es_top;  // [n] the bound m_b of every query; [n x k] exact scores of the approximate top-k (second threshold)
    int32_t cap; bool one_product; float m_abs; const float* q_unscale;
    uint64_t* bmax = nullptr;     // group maxima of the pivot route (one or two queries)
    const float* qsum = nullptr;  // [2 n] sum_i |q_i|, sum_i |q_lo,i| of every query where the query image carries them (launch_query_planes)
    bool m_ready = true;          // hb.m holds the bounds (a threshold kernel ran); false: exact_threshold_kernel computes them from qsum
    bool exact_kth;           // second, tighter threshold from the exact scores of the approximate top-k (RL_OPT_EXACT_KTH_THRESHOLD)
    const uint32_t* filter = nullptr;  // chunk bitset of a metadata-filtered call: masked chunks rank -inf in the approximate scores, so they are
                                       // neither in their top-k nor among the candidates (the bound's argument runs over the chunks that remain)
};
// The scratch layout of a bound-filtered MaxSim batch of n queries (the same in every call that works on the batch)
void hi_batch_layout(rl_index* idx, int32_t n, int32_t k, HiBatch& hb) {
    hb.cap = 2048;  // (the benchmark corpus needs ~300 -- 1 150 with the a-priori bound: score spread sigma ~ 34, window 2 m = 15..34)
    hb.ts = idx->hibuf.as<float>();                                        // [n x k] approximate top-k scores
    hb.ti = reinterpret_cast<int32_t*>(hb.ts + (size_t)n * k);
    hb.thr = reinterpret_cast<float*>(hb.ti + (size_t)n * k);              // [n]
    hb.cnt = reinterpret_cast<uint32_t*>(hb.thr + n);                      // [n]
    hb.flag = hb.cnt + n;                                                  // (16 words)
    hb.ci = reinterpret_cast<int32_t*>(hb.flag + 16);                      // [n x cap] candidate chunks
    hb.es = reinterpret_cast<float*>(hb.ci + (size_t)n * hb.cap);          // [n x cap] their exact scores
    hb.m = hb.es + (size_t)n * hb.cap;                                     // [n]
    hb.es_top = hb.m + n;                                                  // [n x k]
    hb.bmax = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(hb.es_top + (size_t)n * k) + 7) & ~uintptr_t(7));  // [min(n, 2) x 2048] (few-queries pivot route)
}
size_t hi_batch_words(int32_t n, int32_t k) {
    return (size_t)n * k * 3 + (size_t)n * 3 + 16 + (size_t)n * 2048 * 2 + 2 + 2 * pivot_scratch_words(std::min<int32_t>(n, 2));
}
int hi_batch_approx(rl_index* idx, const float* d_q, int32_t nq, int32_t n_queries, int32_t n_gemm, int32_t k, float* sc, int64_t ld, HiBatch& hb,
                    hipStream_t s, bool flag_zeroed = false) {
    // ONE product per multiply -- q_hi.e_hi only, a plain fp16 GEMM -- with the bound widened by what the queries' hi halves drop,
    // (max|e| + max|e_lo|) sum_i |q_lo,i| (measured per query by the threshold kernel).  RL_OPT_HI_PRODUCTS = 2: two products.
    hb.one_product = idx->opt.v[RL_OPT_HI_PRODUCTS] == 1;
    hb.exact_kth = idx->opt.on(RL_OPT_EXACT_KTH_THRESHOLD);
    RL_TRY(idx->hibuf.reserve(hi_batch_words(n_gemm, k) * 4));
    hi_batch_layout(idx, n_gemm, k, hb);
    const int32_t cap = hb.cap;
    // per pair |approx - exact| <= |q_i| |e_lo,j| (what the HI halves drop, measured: max_lo_norm) + 2^-12 |q_i| |e_j| (the
    // query's own 2^-22 split and twice the worst case of a 1024-term fp32 sum, 6e-5)
    hb.m_abs = idx->max_lo_norm + sum_eps(idx->dim) * idx->max_row_norm;
    hb.q_unscale = reinterpret_cast<const float*>(idx->qplanes.as<char>() + (size_t)n_queries * idx->dim * 128);  // launch_query_planes' meta
    hb.qsum = hb.q_unscale + 2 * (size_t)n_queries;                                                                 // ... and its sums
    if (!flag_zeroed) RL_HIP(hipMemsetAsync(hb.flag, 0, 16 * sizeof(uint32_t), s));  // (else: the query-image kernel did, gemm_prepare)
    // (the candidate lists' unused slots -- -1 = "no chunk" -- are filled where the lists are started: hi_batch_rescore)
    (void)cap;
    // One product: SIXTEEN queries per pass through maxsim_pp.hip (dim >= 256; RL_OPT_PP_PASS = 0: the eight-query pass of
    // maxsim_gemm.hip instead -- A/B, and what two products still use).
    const bool pp = hb.one_product && idx->dim >= 256 && idx->opt.on(RL_OPT_PP_PASS);
    // (round 4: ONE launch for all of the batch's passes -- grid row = pass -- so that a pass starts on the CUs the previous one leaves)
    if (pp)
        RL_TRY(launch_maxsim_pp(approx_image(idx), idx->n_rows, idx->dim, idx->qplanes.p, n_queries, 0, n_gemm, nq, idx->row_to_chunk,
                                idx->offsets, idx->ends.as<uint32_t>(), sc, ld, idx->n_cu, s, approx_scale(idx)));
    for (int32_t b = 0; !pp && b < n_gemm; b += GEMM_PASS_QUERIES) {
        const int32_t n_q = std::min<int32_t>(GEMM_PASS_QUERIES, n_gemm - b);
        RL_TRY(launch_maxsim_gemm(approx_image(idx), idx->n_rows, idx->dim, idx->qplanes.p, n_queries, b, n_q, nq, idx->row_to_chunk,
                                  idx->offsets, idx->ends.as<uint32_t>(), sc + (int64_t)b * ld, ld, idx->n_cu, s, approx_scale(idx), true,
                                  nullptr, hb.one_product));
    }
    RL_TRY(mask_chunk_scores(idx, sc, n_gemm, ld, hb.filter, s));  // tombstones (and filtered-out chunks) never become candidates
    // (a handful of queries over more chunks than the one-block selection takes: crowded MaxSim scores send the radix selection down its slow path
    // -- the pivot route first, launch_topk_pivot; it declines what it does not cover)
    int st_pv = RL_ERR_UNSUPPORTED;
    if (n_gemm < 16 && idx->n_chunks > (int64_t)262144 && idx->opt.on(RL_OPT_HI_PIVOT))
        st_pv = launch_topk_pivot(sc, n_gemm, idx->n_chunks, ld, k, idx->ws, hb.ts, hb.ti, s);
    if (st_pv != RL_OK && st_pv != RL_ERR_UNSUPPORTED) return st_pv;
    if (st_pv == RL_ERR_UNSUPPORTED) RL_TRY(launch_topk(sc, n_gemm, idx->n_chunks, ld, k, idx->ws, hb.ts, hb.ti, s));
    return RL_OK;
}
// rows_only: the caller laid out no query fragments (the few-queries route over the HI plane) -- the guarded fallback then streams the rows even
// where the pre-split image exists
int hi_batch_fallback(rl_index* idx, const float* d_q, int32_t nq, int32_t n_queries, int32_t n_gemm, int32_t k, float* sc, int64_t ld,
                      const HiBatch& hb, float* d_s, int32_t* d_c, hipStream_t s, bool rows_only = false);
int hi_batch_rescore(rl_index* idx, const float* d_q, int32_t nq, int32_t n_queries, int32_t n_gemm, int32_t k, float* sc, int64_t ld,
                     const HiBatch& hb, float* d_s, int32_t* d_c, hipStream_t s, bool rows_only = false) {
    const size_t q_elems = (size_t)nq * idx->dim;
    const float* rows = idx->E16 ? reinterpret_cast<const float*>(idx->E16) : idx->E;
    const bool rows16 = idx->E16 != nullptr;
    idx->filt = {RL_FILTER_MAXSIM_BATCH, n_gemm, hb.cap, hb.cnt, hb.flag};
    const int packed = (int)idx->opt.v[RL_OPT_PAIRS_PACKED];
    const bool m_from_qsum = !hb.m_ready && hb.qsum != nullptr;
    if (!hb.m_ready && !(hb.exact_kth && k <= hb.cap && hb.qsum)) return fail(RL_ERR_INVALID, "MaxSim batch: no bound for the candidate threshold");
    if (hb.exact_kth && k <= hb.cap) {
        // Second threshold (hi_filter.hip: exact_threshold_kernel): the approximate top-k is scored exactly FIRST; the k-th best of those
        // exact scores bounds the k-th best overall from below, so a candidate needs approx >= that - m instead of (k-th approx) - 2 m: about
        // half as many chunks beyond the top-k to re-score.  The approximate top-k becomes the head of the list; the collection appends only
        // what ranks below it, and only those entries are scored by the second launch.
        RL_TRY(launch_maxsim_pairs(rows, idx->dim, d_q, nq, (int64_t)q_elems, idx->offsets, hb.ti, k, n_gemm, hb.es_top, s, rows16, 0, 0, packed));
        if (m_from_qsum)  // (also fills the lists' tails with -1)
            RL_TRY(launch_exact_threshold(hb.es_top, hb.ti, n_gemm, k, hb.m, hb.cap, hb.thr, hb.cnt, hb.ci, hb.es, hb.flag, s, hb.qsum, hb.m_abs,
                                          idx->max_row_norm + idx->max_lo_norm, hb.one_product, hb.ts));
        else
            RL_TRY(launch_exact_threshold(hb.es_top, hb.ti, n_gemm, k, hb.m, hb.cap, hb.thr, hb.cnt, hb.ci, hb.es, hb.flag, s, nullptr, 0.f, 0.f, false, hb.ts));
        RL_TRY(launch_collect_above(sc, n_gemm, idx->n_chunks, ld, hb.thr, nullptr, hb.cap, hb.ci, nullptr, hb.cnt, hb.flag, s, hb.ts, hb.ti, k));
        if (hb.cap > k)
            RL_TRY(launch_maxsim_pairs(rows, idx->dim, d_q, nq, (int64_t)q_elems, idx->offsets, hb.ci, hb.cap - k, n_gemm, hb.es, s, rows16, hb.cap, k, packed));
    } else {
        RL_HIP(hipMemsetAsync(hb.ci, 0xff, (size_t)n_gemm * hb.cap * sizeof(int32_t), s));  // unused slots: -1 = "no chunk"
        RL_TRY(launch_collect_above(sc, n_gemm, idx->n_chunks, ld, hb.thr, nullptr, hb.cap, hb.ci, nullptr, hb.cnt, hb.flag, s));
        RL_TRY(launch_maxsim_pairs(rows, idx->dim, d_q, nq, (int64_t)q_elems, idx->offsets, hb.ci, hb.cap, n_gemm, hb.es, s, rows16, 0, 0, packed));
    }
    RL_TRY(launch_merge_topk(hb.es, hb.ci, 1, n_gemm, hb.cap, k, d_s, d_c, s, hb.cnt));
    return hi_batch_fallback(idx, d_q, nq, n_queries, n_gemm, k, sc, ld, hb, d_s, d_c, s, rows_only);
}
int hi_batch_fallback(rl_index* idx, const float* d_q, int32_t nq, int32_t n_queries, int32_t n_gemm, int32_t k, float* sc, int64_t ld,
                      const HiBatch& hb, float* d_s, int32_t* d_c, hipStream_t s, bool rows_only) {
    const size_t q_elems = (size_t)nq * idx->dim;
    // list overflow / unusable bound: the full-precision passes, behind the flag -- ONE launch for all of them (gridDim.y = passes: sixteen
    // guarded launches that return at once were 0.08 ms of every 128-query step)
    // (no pre-split image: the streaming kernels over the rows, one launch with a grid row per query -- the same arithmetic, an order of
    // magnitude slower, and as rare)
    if (image_valid(idx) && !rows_only)
        RL_TRY(launch_maxsim_gemm(idx->planes.p, idx->n_rows, idx->dim, idx->qplanes.p, n_queries, 0, n_gemm, nq, idx->row_to_chunk, idx->offsets,
                                  idx->ends.as<uint32_t>(), sc, ld, idx->n_cu, s, image_scale(idx), idx->E16 != nullptr, hb.flag, false, true));
    else if (idx->dim > 1024)  // (a wide index: no streaming kernel -- the exact re-scoring kernel over EVERY chunk, behind the flag)
        RL_TRY(launch_maxsim_pairs_all_wide(idx->E, idx->dim, d_q, nq, (int64_t)q_elems, idx->offsets, idx->n_chunks, n_gemm, sc, ld, s, hb.flag));
    else  // (grid row = query: with many queries a grid column of n_cu workgroups per query is 32 k workgroups that return at once behind the
          // flag -- 25 us of every step; n_cu / 8 columns keep the device as full when the passes do run and cost 3 us when they do not)
        RL_TRY(launch_maxsim_stream_batch(idx->E, false, idx->n_rows, idx->dim, d_q, nq, (int64_t)q_elems, n_gemm, idx->row_to_chunk, idx->offsets,
                                          idx->n_chunks, sc, ld, std::max(1, idx->n_cu / std::min<int32_t>(8, std::max<int32_t>(1, n_gemm / 4))), s,
                                          idx->split_scale, hb.flag));
    RL_TRY(mask_chunk_scores(idx, sc, n_gemm, ld, hb.filter, s));  // (harmless on scores nobody reads)
    // (the exact top-k of the fallback's scores in ONE guarded launch, a block per query -- select.hip: guarded_select_kernel -- instead of the
    // selection's three: what usually returns at once is one launch shorter by two)
    uint32_t* host_word = nullptr;
    if (idx->opt.on(RL_OPT_LAZY_IMAGES) && !image_valid(idx) && !rows_only) {  // (lazy images: let the next batch know whether this one fell back)
        if (!idx->h_fell_back) {  // (no pinned word: no signal -- the fallback then stays on the streaming kernels, results unchanged)
            if (hipHostMalloc(reinterpret_cast<void**>(&idx->h_fell_back), sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) {
                *idx->h_fell_back = 0u;
                if (hipHostGetDevicePointer(reinterpret_cast<void**>(&idx->d_fell_back), idx->h_fell_back, 0) != hipSuccess) {
                    idx->d_fell_back = nullptr;
                    (void)hipGetLastError();
                }
            } else { idx->h_fell_back = nullptr; (void)hipGetLastError(); }
        }
        if (idx->h_fell_back) host_word = idx->d_fell_back;
        // (no device view of the pinned word: a 4-byte copy behind the selection, as before round 6)
        if (idx->h_fell_back && !host_word) {
            RL_TRY(launch_guarded_select(sc, n_gemm, idx->n_chunks, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, d_s, d_c, hb.flag, s));
            RL_HIP(hipMemcpyAsync(idx->h_fell_back, hb.flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            return RL_OK;
        }
    }
    RL_TRY(launch_guarded_select(sc, n_gemm, idx->n_chunks, ld, k, nullptr, nullptr, nullptr, 0, SCAN_RAW_DOT, 1.0f, d_s, d_c, hb.flag, s, host_word));
    return RL_OK;
}

// ---- one or two MaxSim queries over an fp32 index in split arithmetic: the half-width route (round 6) ------------------------------------
// How the reference calls the reranker -- one user query at a time (src/raglite/_search.py:394-396) -- used to stream the fp32 rows (4 B per
// element: maxsim_stream_kernel / maxsim_stream2_kernel, 0.58-0.67 ms per pass at 1 M x 1024): the bound-filtered pipeline of the batches
// starts at three queries because its sixteen-query pass is matrix-pipe-bound whatever the number of queries.  For one or two queries the
// APPROXIMATE pass is the HBM-bound streaming kernel over the row-major fp16 HI plane (2 B per element: the image the single-query row search
// ranks from) -- both halves of the query multiplied, so all it drops is e_lo -- and the rest is the batch's own pipeline: exact top-k of the
// approximate scores, |approximate - exact| <= m = (max|e_lo| + 2^-12 max|e|) sum_i |q_i|, second threshold from the exact scores of the
// approximate top-k, exact re-scoring of the candidates over the rows (maxsim_pairs_kernel), ranking; list overflow / unusable bound ->
// device flag -> the streaming pass over the rows + exact selection behind it.  Results: the exact top-k of exactly computed scores.
// d_q: the n (1 or 2) queries, q_elems floats apart; sc: [n x ld] scratch rows; d_s / d_c: [n x k].  RL_ERR_UNSUPPORTED where the route
// does not apply (the caller keeps the streaming kernels over the rows).
int maxsim_few_hi_plane(rl_index* idx, const float* d_q, int32_t nq, int32_t n, int32_t k, float* sc, int64_t ld, float* d_s, int32_t* d_c,
                        hipStream_t s, const uint32_t* d_filter) {
    if (n < 1 || n > 2 || nq < 1 || nq > 32 || k > 512 || !idx->opt.on(RL_OPT_HI_MAXSIM) || !idx->opt.on(RL_OPT_HI_FEW)) return RL_ERR_UNSUPPORTED;
    if (idx->dim > 1024) return RL_ERR_UNSUPPORTED;  // (the stream kernels' dims: a wide index takes the pass even for one query, gemm_min_queries)
    if (idx->E16 || !idx->E || !(idx->split_scale > 0.f) || idx->has_empty_chunk || idx->n_chunks == 0 || idx->n_rows == 0) return RL_ERR_UNSUPPORTED;
    RL_TRY(demand_images(idx, IMG_HI_PLANE, s));
    if (!hi_valid(idx) || idx->max_row_norm_rows != idx->n_rows || !(idx->max_row_norm > 0.f)) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(d_q) & 15) || (reinterpret_cast<uintptr_t>(idx->E) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t q_elems = (int64_t)nq * idx->dim;
    int st = RL_ERR_UNSUPPORTED;
    if (n == 2 && nq > 16 && idx->opt.on(RL_OPT_QUERY_PAIRS)) {  // two queries of 17..32 vectors share ONE pass over the plane
        RL_TRY(idx->qsplit.reserve(query_split_bytes(idx->dim, 2)));
        st = launch_query_split(d_q, idx->dim, nq, q_elems, 2, idx->qsplit.as<char>(), true, s);
        if (st == RL_OK)
            st = launch_maxsim_stream2(idx->hiplane.p, true, idx->n_rows, idx->dim, idx->qsplit.as<char>(), 2, 0, nq, idx->row_to_chunk, idx->offsets,
                                       idx->n_chunks, sc, ld, idx->n_cu, s, 1.f);
    }
    if (st == RL_ERR_UNSUPPORTED)  // one pass per query, one launch (grid row = query)
        st = launch_maxsim_stream_batch(idx->hiplane.p, true, idx->n_rows, idx->dim, d_q, nq, q_elems, n, idx->row_to_chunk, idx->offsets,
                                        idx->n_chunks, sc, ld, idx->n_cu, s, 0.f, nullptr);
    if (st != RL_OK) return st;
    HiBatch hb;
    hb.one_product = false;  // q_hi . e_hi + q_lo . e_hi: nothing of the query is dropped
    hb.exact_kth = idx->opt.on(RL_OPT_EXACT_KTH_THRESHOLD);
    RL_TRY(idx->hibuf.reserve(hi_batch_words(n, k) * 4));
    hi_batch_layout(idx, n, k, hb);
    hb.m_abs = idx->max_lo_norm + sum_eps(idx->dim) * idx->max_row_norm;
    hb.q_unscale = nullptr;
    hb.filter = d_filter;
    // The pivot route (round 6, option hi_pivot; k <= 128, no tombstones): with the chip idle around one query, re-scoring ~2.5 x the candidates
    // costs nothing, so the approximate scores need not be RANKED -- no top-k, no second threshold from its exact scores: the k-th largest of
    // ~490 wave maxima bounds the k-th best from below, every chunk within 2 m of it is re-scored in ONE launch and ranked.  Four launches
    // behind the pass (maxima + scale + bound, pivot + collection, exact scores, ranking) instead of nine.
    if (idx->opt.on(RL_OPT_HI_PIVOT) && !idx->live_chunk_bits && !d_filter && k <= 128) {
        HiBound bound;
        bound.m_out = hb.m;
        PivotMaxSim ms;
        ms.nq = nq; ms.q_stride = q_elems; ms.m_abs = hb.m_abs; ms.fill_ids = hb.ci;
        const int st_pv = launch_pivot_route(sc, n, idx->n_chunks, ld, k, nullptr, nullptr, d_q, idx->dim, SCAN_RAW_DOT, 1.0f / idx->split_scale, hb.bmax,
                                             hb.cnt, n + 16, bound, hb.thr, hb.cap, hb.ci, nullptr, hb.cnt, hb.flag, s, nullptr, nullptr, nullptr, &ms);
        if (st_pv == RL_OK) {
            idx->filt = {RL_FILTER_MAXSIM_BATCH, n, hb.cap, hb.cnt, hb.flag};
            // (one candidate per wave here: the eight-wave kernel's 256-k blocks -- half as many dependent HBM round trips per tile -- suit it better
            // than the sixteen-wave variant's 128-k blocks, which pay when a wave has many tiles to pipeline)
            RL_TRY(launch_maxsim_pairs(idx->E, idx->dim, d_q, nq, q_elems, idx->offsets, hb.ci, hb.cap, n, hb.es, s, false, 0, 0,
                                       std::min<int>(1, (int)idx->opt.v[RL_OPT_PAIRS_PACKED])));
            RL_TRY(launch_merge_topk(hb.es, hb.ci, 1, n, hb.cap, k, d_s, d_c, s, hb.cnt));
            return hi_batch_fallback(idx, d_q, nq, n, n, k, sc, ld, hb, d_s, d_c, s, true);
        }
        if (st_pv != RL_ERR_UNSUPPORTED) return st_pv;
    }
    // the plane holds fp16(e * scale), scale a power of two: undone exactly (the launch also zeroes the flag block: the two small memsets of
    // this route were six fill launches, 28 us of a 0.44 ms query)
    RL_TRY(launch_scale_f32(sc, sc, 1.0f / idx->split_scale, (int64_t)(n - 1) * ld + idx->n_chunks, s, hb.flag, 16));
    RL_TRY(mask_chunk_scores(idx, sc, n, ld, d_filter, s));  // tombstones / filtered-out chunks never become candidates
    RL_TRY(launch_topk(sc, n, idx->n_chunks, ld, k, idx->ws, hb.ts, hb.ti, s));
    RL_TRY(launch_maxsim_threshold(hb.ts, n, k, d_q, nq, idx->dim, q_elems, 1.0f, hb.m_abs, hb.thr, hb.cnt, hb.flag, s, nullptr,
                                   idx->max_row_norm + idx->max_lo_norm, hb.m));
    return hi_batch_rescore(idx, d_q, nq, n, n, k, sc, ld, hb, d_s, d_c, s, true);
}
// The device side of rl_maxsim_topk_batch: d_q [n_queries x nq x dim] -> d_s / d_c [n_queries x k]; sc: [n_queries x ld] scratch rows
// (idx->scores); q16: the queries were widened from fp16 (exact).  Also what rl_maxsim_topk runs on a WIDE index, as a batch of one.
int maxsim_topk_batch_device(rl_index* idx, const float* d_q, bool q16, int32_t n_queries, int32_t nq, int32_t k, float* sc, int64_t ld, float* d_s,
                             int32_t* d_c, hipStream_t s, const uint32_t* d_filter) {
    const size_t q_elems = (size_t)nq * idx->dim;
    // One corpus pass per query, back to back on the caller's stream.  (Round-robin over side streams hides each
    // launch's tail behind the next one's ramp and measured +1.5 % at 1 M rows / +3.3 % on a 125 k-row shard, but
    // concurrent kernels stretch each other's durations 3x in a kernel trace, which would make the rocprofv3 summary
    // disagree with the live roofline timing; the serial form keeps every number checkable.)
    // Two queries share a pass where the arithmetic allows it (fp16-split, 17..32 vectors per query); an odd query out,
    // and every query otherwise, takes a pass of its own.
    // Eight queries share a pass over the pre-split corpus image where the index has one (maxsim_gemm.hip); what is left
    // of the batch (fewer than three queries) goes through the streaming kernels as before.
    int32_t base = 0;
    bool hi_done = false;  // queries [0, base) were ranked by the half-bytes pipeline below (results already in d_s / d_c)
    {
        const int32_t n_gemm = n_queries - base >= gemm_min_queries(idx)
                                   ? (n_queries / GEMM_PASS_QUERIES) * GEMM_PASS_QUERIES +
                                         ((n_queries % GEMM_PASS_QUERIES) >= gemm_min_queries(idx) ? n_queries % GEMM_PASS_QUERIES : 0)
                                   : 0;
        // (the flag block of the bound-filtered pipeline lies in a pool sized by n_gemm and k alone: reserved here so that the query-image
        // kernel can zero it on its way -- one memset launch less per batch)
        uint32_t* flag_words = nullptr;
        if (n_gemm > 0 && k <= 512) {
            RL_TRY(idx->hibuf.reserve(hi_batch_words(n_gemm, k) * 4));
            HiBatch pre;
            hi_batch_layout(idx, n_gemm, k, pre);
            flag_words = pre.flag;
        }
        const int st = gemm_prepare(idx, d_q, nq, (int64_t)q_elems, n_queries, s, k > 512, flag_words);
        if (st == RL_OK) {
            const bool hi_off = !idx->opt.on(RL_OPT_HI_MAXSIM);
            const bool two_products = idx->opt.v[RL_OPT_HI_PRODUCTS] == 2;  // (over an fp16-stored corpus two products ARE the full precision)
            const bool pp_route = idx->opt.v[RL_OPT_HI_PRODUCTS] == 1 && idx->dim >= 256 && idx->opt.on(RL_OPT_PP_PASS);
            const bool slim = !image_valid(idx);  // (gemm_prepare accepted it: slim_batch_ok) -- only the sixteen-query pass reads the HI image alone
            if (n_gemm > 0 && !hi_off && approx_image_valid(idx) && k <= 512 && !(idx->E16 && two_products) && (!slim || pp_route)) {
                // ---- MaxSim of a batch at two MFMA products per multiply instead of three (the headline path) -------------------------
                // (1) approximate chunk scores: the eight-query pass over the HI image (q_hi.e_hi + q_lo.e_hi);
                // (2) |approximate - exact| <= m = (max|e_lo| + 2^-12 max|e|) sum_i |q_i| for every chunk (the per-pair bound of
                //     search_rows_hi under the max over a chunk's rows and the sum over the query vectors), so the chunks with
                //     approximate score >= (k-th best approximate) - 2 m contain the exact top-k: collected per query;
                // (3) their exact scores by maxsim_pairs_kernel (fp32 matrix pipe), ranked by (score desc, chunk asc);
                // (4) list overflow / unusable bound -> device flag -> the full-precision passes + selection, launched always,
                //     returning at once when the flag is clear.
                // ONE product per multiply -- q_hi.e_hi only, a plain fp16 GEMM -- with the bound widened by what the queries' hi
                // halves drop, (max|e| + max|e_lo|) sum_i |q_lo,i| (measured per query by the threshold kernel): same results by the
                // same argument (bit-identical on the benchmark shape, profiles/r02_u_probe.txt), the pass 1.01 -> 0.71 ms, 418 instead of
                // 307 candidates per query, the 128-query step 17.1 -> 12.8 ms.  RL_OPT_HI_PRODUCTS = 2: two products.
                HiBatch hb;
                hb.filter = d_filter;
                RL_TRY(hi_batch_approx(idx, d_q, nq, n_queries, n_gemm, k, sc, ld, hb, s, flag_words != nullptr));
                // (an fp32-STORED corpus whose every element is an fp16 value at the split scale -- what its HI halves drop was measured as
                // exactly zero when the image was built -- is the same case: RAGLite's embeddings handed over as float32 arrays)
                if (q16 && (idx->E16 || idx->max_lo_norm == 0.f) && hb.one_product && idx->opt.on(RL_OPT_F16_EXACT)) {
                    // fp16 queries x fp16-stored corpus: q_hi . e IS q . e (products of two fp16 values are exact in fp32; the sums are the
                    // pass's fp32 accumulation) -- its top-k is the result; certified per query, the full-precision passes behind the flag
                    RL_TRY(launch_f16_exact_finish(d_q, nq, idx->dim, (int64_t)q_elems, hb.q_unscale, hb.ts, hb.ti, n_gemm, k, d_s, d_c, hb.cnt, hb.flag, s));
                    RL_TRY(hi_batch_fallback(idx, d_q, nq, n_queries, n_gemm, k, sc, ld, hb, d_s, d_c, s));
                    idx->filt = {RL_FILTER_MAXSIM_F16_EXACT, n_gemm, hb.cap, hb.cnt, hb.flag};
                } else {
                    // (the second threshold needs the bound m_b only: exact_threshold_kernel takes it from the sums the query-image kernel left
                    // -- no threshold kernel that reads the queries again: 10 us of every 128-query step)
                    if (hb.exact_kth && k <= hb.cap && hb.qsum) hb.m_ready = false;
                    else RL_TRY(launch_maxsim_threshold(hb.ts, n_gemm, k, d_q, nq, idx->dim, (int64_t)q_elems, 1.0f, hb.m_abs, hb.thr, hb.cnt, hb.flag, s,
                                                        hb.one_product ? hb.q_unscale : nullptr, idx->max_row_norm + idx->max_lo_norm, hb.m));
                    RL_TRY(hi_batch_rescore(idx, d_q, nq, n_queries, n_gemm, k, sc, ld, hb, d_s, d_c, s));
                }
                base = n_gemm;
                hi_done = true;
            } else if (!slim) {
                while (n_queries - base >= gemm_min_queries(idx)) {
                    const int32_t n_q = std::min<int32_t>(GEMM_PASS_QUERIES, n_queries - base);
                    RL_TRY(gemm_pass(idx, nq, n_queries, base, n_q, sc + (int64_t)base * ld, ld, s));
                    base += n_q;
                }
            }
        } else if (st != RL_ERR_UNSUPPORTED) {
            return st;
        }
    }
    int32_t rest = n_queries - base;
    int32_t few = 0;  // queries [base, base + few) were ranked by the few-queries route over the HI plane
    if (rest >= 1 && rest <= 2) {
        const int st = maxsim_few_hi_plane(idx, d_q + (size_t)base * q_elems, nq, rest, k, sc + (int64_t)base * ld, ld, d_s + (int64_t)base * k,
                                           d_c + (int64_t)base * k, s, d_filter);
        if (st == RL_OK) { few = rest; rest = 0; }
        else if (st != RL_ERR_UNSUPPORTED) return st;
    }
    int32_t paired = 0;
    if (rest >= 2) {
        const int st = pairs_prepare(idx, d_q + (size_t)base * q_elems, nq, (int64_t)q_elems, rest & ~1, s);
        if (st == RL_OK) paired = rest & ~1;
        else if (st != RL_ERR_UNSUPPORTED) return st;
    }
    for (int32_t b = 0; b < paired; b += 2) RL_TRY(pairs_pass(idx, nq, paired, b, sc + (int64_t)(base + b) * ld, ld, s));
    for (int32_t b = base + paired; b < n_queries - few; ++b)
        RL_TRY(maxsim_scores_device(idx, d_q + (size_t)b * q_elems, nq, sc + (int64_t)b * ld, s));
    {   // what the half-bytes pipelines did not rank: every query, or the one or two left over
        const int32_t first = few ? n_queries : (hi_done ? base : 0);
        if (n_queries > first) {
            RL_TRY(mask_chunk_scores(idx, sc + (int64_t)first * ld, n_queries - first, ld, d_filter, s));  // tombstones / the call's filter
            RL_TRY(launch_topk(sc + (int64_t)first * ld, n_queries - first, idx->n_chunks, ld, k, idx->ws, d_s + (int64_t)first * k,
                               d_c + (int64_t)first * k, s));
        }
    }
    if (idx->live_chunk_bits || d_filter) RL_TRY(launch_fix_masked(d_s, d_c, (int64_t)n_queries * k, s));
    return RL_OK;
}
}  // namespace

// query_vecs: [n_queries x nq x dim] fp32 -- or, q16, IEEE fp16 (rl_maxsim_topk_batch_f16): widened on the device (exact), and over an
// fp16-stored index the one-product pass then IS the score (hi_filter.hip: f16_exact_finish_kernel)
static int maxsim_topk_batch_any(rl_index* idx, const void* query_vecs, bool q16, int32_t n_queries, int32_t nq, int32_t k,
                                 float* out_scores, int32_t* out_chunks, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_topk_batch: null index");
    if (n_queries < 0 || nq < 1) return fail(RL_ERR_INVALID, "rl_maxsim_topk_batch: bad sizes");
    if (k < 1) return fail(RL_ERR_INVALID, "rl_maxsim_topk_batch: k must be >= 1");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_topk_batch: k must be <= 2048");
    if (n_queries == 0) return RL_OK;
    if (!query_vecs || !out_scores || !out_chunks) return fail(RL_ERR_INVALID, "rl_maxsim_topk_batch: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_s, t_c;
    const float* d_q; float* d_s; int32_t* d_c;
    const size_t q_elems = (size_t)nq * idx->dim;
    if (q16) {
        const uint16_t* d_q16;
        RL_TRY(stage_in(static_cast<const uint16_t*>(query_vecs), (size_t)n_queries * q_elems, mem, s, t_q, &d_q16));
        RL_TRY(idx->q32.reserve((size_t)n_queries * q_elems * sizeof(float)));
        const int st_w = launch_widen_f16(d_q16, idx->q32.as<float>(), (int64_t)n_queries * (int64_t)q_elems, s);
        if (st_w == RL_ERR_UNSUPPORTED) return fail(RL_ERR_INVALID, "rl_maxsim_topk_batch_f16: query_vecs_f16 must be 2-byte aligned");
        RL_TRY(st_w);  // (a launch failure stays what it is)
        d_q = idx->q32.as<float>();
    } else {
        RL_TRY(stage_in(static_cast<const float*>(query_vecs), (size_t)n_queries * q_elems, mem, s, t_q, &d_q));
    }
    RL_TRY(stage_out_begin(out_scores, (size_t)n_queries * k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_chunks, (size_t)n_queries * k, mem, t_c, &d_c));
    const int64_t ld = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
    RL_TRY(idx->scores.reserve((size_t)n_queries * ld * sizeof(float)));
    float* sc = idx->scores.as<float>();
    RL_TRY(maxsim_topk_batch_device(idx, d_q, q16, n_queries, nq, k, sc, ld, d_s, d_c, s));
    RL_TRY(stage_out_end(out_scores, (size_t)n_queries * k, mem, s, t_s));
    RL_TRY(stage_out_end(out_chunks, (size_t)n_queries * k, mem, s, t_c));
    return finish(mem, s);
}

int rl_maxsim_topk_batch(rl_index* idx, const float* query_vecs, int32_t n_queries, int32_t nq, int32_t k,
                         float* out_scores, int32_t* out_chunks, int mem, void* stream) {
    return maxsim_topk_batch_any(idx, query_vecs, false, n_queries, nq, k, out_scores, out_chunks, mem, stream);
}

int rl_maxsim_topk_batch_f16(rl_index* idx, const uint16_t* query_vecs_f16, int32_t n_queries, int32_t nq, int32_t k,
                             float* out_scores, int32_t* out_chunks, int mem, void* stream) {
    return maxsim_topk_batch_any(idx, query_vecs_f16, true, n_queries, nq, k, out_scores, out_chunks, mem, stream);
}

// ---- the MaxSim batch of a corpus SHARDED over several indexes, with ONE candidate threshold for all shards (header: protocol) ------------
int rl_maxsim_batch_begin(rl_index* idx, const float* query_vecs, int32_t n_queries, int32_t nq, int32_t k, float* out_approx, int mem,
                          void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_batch_begin: null index");
    if (n_queries < 1 || nq < 1 || k < 1) return fail(RL_ERR_INVALID, "rl_maxsim_batch_begin: bad sizes");
    if (!query_vecs || !out_approx) return fail(RL_ERR_INVALID, "rl_maxsim_batch_begin: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    idx->mb_B = 0;
    // the bound-filtered pipeline must cover the WHOLE batch (rl_maxsim_topk_batch takes other kernels for what it leaves over)
    const bool hi_off = !idx->opt.on(RL_OPT_HI_MAXSIM);
    const bool whole = n_queries % GEMM_PASS_QUERIES == 0 || n_queries % GEMM_PASS_QUERIES >= gemm_min_queries(idx);
    if (hi_off || !whole || n_queries < gemm_min_queries(idx) || k > 512 || nq > 32)
        return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_batch_begin: needs >= 3 queries (n % 8 == 0 or n % 8 >= 3), nq <= 32 and k <= 512");
    DevBuf t_q, t_o;
    const float* d_q; float* d_o;
    const size_t q_elems = (size_t)nq * idx->dim;
    RL_TRY(stage_in(query_vecs, (size_t)n_queries * q_elems, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_approx, (size_t)n_queries * (k + 1), mem, t_o, &d_o));
    const int64_t ld = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
    RL_TRY(idx->scores.reserve((size_t)n_queries * ld * sizeof(float)));
    float* sc = idx->scores.as<float>();
    {
        const int st = gemm_prepare(idx, d_q, nq, (int64_t)q_elems, n_queries, s);
        if (st != RL_OK) return st == RL_ERR_UNSUPPORTED ? fail(RL_ERR_UNSUPPORTED, "rl_maxsim_batch_begin: this index keeps no corpus image") : st;
    }
    if (!approx_image_valid(idx)) return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_batch_begin: this index keeps no image for the approximate pass (small or exact-fp32 index)");
    if (!image_valid(idx) && !(idx->opt.v[RL_OPT_HI_PRODUCTS] == 1 && idx->dim >= 256 && idx->opt.on(RL_OPT_PP_PASS)))
        return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_batch_begin: without the pre-split image only the sixteen-query pass (hi_products = 1, pp_pass = 1) runs");
    HiBatch hb;
    RL_TRY(hi_batch_approx(idx, d_q, nq, n_queries, n_queries, k, sc, ld, hb, s));
    // this shard's bound m_b: the threshold kernel over a "k-th best" of zero leaves -2 m_b
    RL_HIP(hipMemsetAsync(hb.es, 0, (size_t)n_queries * sizeof(float), s));
    RL_TRY(launch_maxsim_threshold(hb.es, n_queries, 1, d_q, nq, idx->dim, (int64_t)q_elems, 1.0f, hb.m_abs, hb.thr, hb.cnt, hb.flag, s,
                                   hb.one_product ? hb.q_unscale : nullptr, idx->max_row_norm + idx->max_lo_norm));
    RL_TRY(launch_pack_approx(hb.ts, hb.thr, n_queries, k, d_o, s));
    idx->mb_B = n_queries;
    idx->mb_nq = nq;
    idx->mb_k = k;
    idx->mb_epoch = idx->scratch_epoch;
    RL_TRY(stage_out_end(out_approx, (size_t)n_queries * (k + 1), mem, s, t_o));
    return finish(mem, s);
}

int rl_maxsim_batch_finish(rl_index* idx, const float* query_vecs, const float* all_approx, int32_t world, int32_t rank, float* out_scores,
                           int32_t* out_chunks, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_batch_finish: null index");
    if (!query_vecs || !all_approx || !out_scores || !out_chunks) return fail(RL_ERR_INVALID, "rl_maxsim_batch_finish: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(RL_ERR_INVALID, "rl_maxsim_batch_finish: bad world / rank");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    if (idx->mb_B < 1) return fail(RL_ERR_INVALID, "rl_maxsim_batch_finish: no rl_maxsim_batch_begin in progress on this index");
    if (idx->scratch_epoch != idx->mb_epoch + 1) {  // (this call's own use_scratch is the + 1)
        idx->mb_B = 0;
        return fail(RL_ERR_INVALID, "rl_maxsim_batch_finish: another call used this index since rl_maxsim_batch_begin (its approximate scores are gone)");
    }
    const int32_t n_queries = idx->mb_B, nq = idx->mb_nq, k = idx->mb_k;
    idx->mb_B = 0;
    DevBuf t_q, t_a, t_s, t_c;
    const float* d_q; const float* d_a; float* d_s; int32_t* d_c;
    const size_t q_elems = (size_t)nq * idx->dim;
    RL_TRY(stage_in(query_vecs, (size_t)n_queries * q_elems, mem, s, t_q, &d_q));
    RL_TRY(stage_in(all_approx, (size_t)world * n_queries * (k + 1), mem, s, t_a, &d_a));
    RL_TRY(stage_out_begin(out_scores, (size_t)n_queries * k, mem, t_s, &d_s));
    RL_TRY(stage_out_begin(out_chunks, (size_t)n_queries * k, mem, t_c, &d_c));
    const int64_t ld = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
    float* sc = idx->scores.as<float>();
    HiBatch hb;  // the layout of rl_maxsim_batch_begin (same sizes: nothing is reallocated, the approximate scores are still in `sc`)
    hi_batch_layout(idx, n_queries, k, hb);
    hb.one_product = idx->opt.v[RL_OPT_HI_PRODUCTS] == 1;
    hb.exact_kth = false;  // (the shards' ONE threshold comes from the exchange of their approximate lists)
    hb.m_abs = 0.f;
    hb.q_unscale = nullptr;
    RL_TRY(launch_global_threshold(d_a, world, n_queries, k, rank, hb.thr, hb.cnt, hb.flag, s));
    RL_TRY(hi_batch_rescore(idx, d_q, nq, n_queries, n_queries, k, sc, ld, hb, d_s, d_c, s));
    if (idx->live_chunk_bits) RL_TRY(launch_fix_masked(d_s, d_c, (int64_t)n_queries * k, s));
    RL_TRY(stage_out_end(out_scores, (size_t)n_queries * k, mem, s, t_s));
    RL_TRY(stage_out_end(out_chunks, (size_t)n_queries * k, mem, s, t_c));
    return finish(mem, s);
}

int rl_maxsim_approx_scores(rl_index* idx, const float* query_vecs, int32_t n_queries, int32_t nq, int kernel, float* out_scores,
                            float* out_bound, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_approx_scores: null index");
    if (n_queries < 0 || nq < 1 || (kernel != 0 && kernel != 1)) return fail(RL_ERR_INVALID, "rl_maxsim_approx_scores: bad arguments");
    if (n_queries == 0) return RL_OK;
    if (!query_vecs || !out_scores) return fail(RL_ERR_INVALID, "rl_maxsim_approx_scores: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    // (both pass kernels find a chunk by counting chunk ends: an empty chunk would shift every score behind it to the wrong chunk)
    if (idx->has_empty_chunk || idx->n_chunks == 0)
        return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_approx_scores: the approximate pass needs an index without empty chunks");
    RL_TRY(demand_images(idx, approx_image_bit(idx), s));
    if (!approx_image_valid(idx) || nq > 32 || (kernel == 0 && idx->dim < 256))
        return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_approx_scores: this index keeps no HI image (or nq > 32 / dim < 256)");
    DevBuf t_q, t_o, t_b;
    const float* d_q; float* d_o; float* d_b = nullptr;
    const size_t q_elems = (size_t)nq * idx->dim;
    const int64_t ld = idx->n_chunks;
    RL_TRY(stage_in(query_vecs, (size_t)n_queries * q_elems, mem, s, t_q, &d_q));
    RL_TRY(stage_out_begin(out_scores, (size_t)n_queries * ld, mem, t_o, &d_o));
    if (out_bound) RL_TRY(stage_out_begin(out_bound, (size_t)n_queries, mem, t_b, &d_b));
    RL_TRY(idx->qplanes.reserve(query_planes_bytes(idx->dim, n_queries)));
    RL_TRY(launch_query_planes(d_q, idx->dim, nq, (int64_t)q_elems, n_queries, idx->qplanes.p, s));
    const int32_t per = kernel == 0 ? n_queries : GEMM_PASS_QUERIES;  // (the sixteen-query kernel takes all its passes in one launch)
    for (int32_t b = 0; b < n_queries; b += per) {
        const int32_t n_q = std::min<int32_t>(per, n_queries - b);
        if (kernel == 0)
            RL_TRY(launch_maxsim_pp(approx_image(idx), idx->n_rows, idx->dim, idx->qplanes.p, n_queries, b, n_q, nq, idx->row_to_chunk, idx->offsets,
                                    idx->ends.as<uint32_t>(), d_o + (int64_t)b * ld, ld, idx->n_cu, s, approx_scale(idx)));
        else
            RL_TRY(launch_maxsim_gemm(approx_image(idx), idx->n_rows, idx->dim, idx->qplanes.p, n_queries, b, n_q, nq, idx->row_to_chunk, idx->offsets,
                                      idx->ends.as<uint32_t>(), d_o + (int64_t)b * ld, ld, idx->n_cu, s, approx_scale(idx), true, nullptr, true));
    }
    if (d_b) {  // the bound of the one-product pass, by the kernel the pipeline computes its thresholds with (k = 1 over a dummy top list)
        const float m_abs = idx->max_lo_norm + sum_eps(idx->dim) * idx->max_row_norm;
        const float* q_unscale = reinterpret_cast<const float*>(idx->qplanes.as<char>() + (size_t)n_queries * idx->dim * 128);
        RL_TRY(idx->hibuf.reserve((size_t)n_queries * 4 * sizeof(float) + 64));
        float* top = idx->hibuf.as<float>();                                     // [n] "k-th best" = 0
        float* thr = top + n_queries;                                            // [n] 0 - 2 m
        uint32_t* cnt = reinterpret_cast<uint32_t*>(thr + n_queries);            // [n]
        uint32_t* flag = cnt + n_queries;
        RL_HIP(hipMemsetAsync(top, 0, (size_t)n_queries * sizeof(float), s));
        RL_HIP(hipMemsetAsync(flag, 0, 16 * sizeof(uint32_t), s));
        RL_TRY(launch_maxsim_threshold(top, n_queries, 1, d_q, nq, idx->dim, (int64_t)q_elems, 1.0f, m_abs, thr, cnt, flag, s, q_unscale,
                                       idx->max_row_norm + idx->max_lo_norm));
        RL_TRY(launch_scale_f32(thr, d_b, -0.5f, n_queries, s));                // thr = -2 m  ->  m
        idx->filt = {};
    }
    RL_TRY(stage_out_end(out_scores, (size_t)n_queries * ld, mem, s, t_o));
    if (out_bound) RL_TRY(stage_out_end(out_bound, (size_t)n_queries, mem, s, t_b));
    return finish(mem, s);
}

int rl_maxsim_rerank(rl_index* idx, const float* query_vecs, int32_t n_queries, int32_t nq, const int32_t* candidates,
                     int32_t n_cand, float* out_scores, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_maxsim_rerank: null index");
    if (n_queries < 0 || n_cand < 0 || nq < 1) return fail(RL_ERR_INVALID, "rl_maxsim_rerank: bad sizes");
    if (n_queries == 0 || n_cand == 0) return RL_OK;
    if (!query_vecs || !candidates || !out_scores) return fail(RL_ERR_INVALID, "rl_maxsim_rerank: null argument");
    if (idx->E16 && (nq > 32 || idx->dim % 16))
        return fail(RL_ERR_UNSUPPORTED, "rl_maxsim_rerank on an fp16-stored index needs nq <= 32 and dim % 16 == 0");
    if (mem == RL_MEM_HOST) {  // -1 = "no chunk" (the padding of rl_search_chunks results) is allowed and scores -inf
        for (int64_t i = 0; i < (int64_t)n_queries * n_cand; ++i)
            if (candidates[i] < -1 || candidates[i] >= idx->n_chunks)
                return fail(RL_ERR_INVALID, "rl_maxsim_rerank: candidate chunk ordinal out of range");
    }
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_c, t_o;
    const float* d_q; const int32_t* d_c; float* d_o;
    RL_TRY(stage_in(query_vecs, (size_t)n_queries * nq * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_in(candidates, (size_t)n_queries * n_cand, mem, s, t_c, &d_c));
    RL_TRY(stage_out_begin(out_scores, (size_t)n_queries * n_cand, mem, t_o, &d_o));
    {   // ordinals that cannot be scored (device callers are not validated above; -1 pads of a search result; tombstones) -> -inf
        const int64_t n_items = (int64_t)n_queries * n_cand;
        RL_TRY(idx->cand.reserve((size_t)n_items * sizeof(int32_t)));
        RL_TRY(launch_sanitize_candidates(d_c, n_items, idx->n_chunks, idx->live_chunk_bits, idx->cand.as<int32_t>(), s));
        d_c = idx->cand.as<int32_t>();
    }
    int st = idx->E16 ? launch_maxsim_cand16(idx->E16, idx->dim, d_q, nq, idx->offsets, d_c, n_cand, n_queries, d_o, s)
                      : launch_maxsim_cand(idx->E, idx->dim, d_q, nq, idx->offsets, d_c, n_cand, n_queries, d_o, s, idx->split_scale);
    if (st == RL_ERR_UNSUPPORTED)  // other dims: the fp32-MFMA pairs kernels (dim % 16 == 0 up to 1024, % 128 up to 4096, nq <= 32; fp16 rows widened on the way in) ...
        st = launch_maxsim_pairs(idx->E16 ? reinterpret_cast<const float*>(idx->E16) : idx->E, idx->dim, d_q, nq, (int64_t)nq * idx->dim, idx->offsets, d_c,
                                 n_cand, n_queries, d_o, s, idx->E16 != nullptr, 0, 0, (int)idx->opt.v[RL_OPT_PAIRS_PACKED]);
    if (st == RL_ERR_UNSUPPORTED && idx->E16) return fail(st, "rl_maxsim_rerank: unsupported shape for an fp16-stored index");
    if (st == RL_ERR_UNSUPPORTED && !idx->E16)  // ... and the VALU backstop for everything else
        st = launch_maxsim_generic(idx->E, idx->dim, d_q, nq, (int64_t)nq * idx->dim, idx->offsets, d_c, n_cand,
                                   n_queries, d_o, s);
    RL_TRY(st);
    RL_TRY(stage_out_end(out_scores, (size_t)n_queries * n_cand, mem, s, t_o));
    return finish(mem, s);
}

// ---- section 8e + generic selection ----------------------------------------------------------------------
int rl_merge_topk(const float* in_scores, const int32_t* in_ids, int32_t n_lists, int32_t n_queries, int32_t k_in,
                  int32_t k, float* out_scores, int32_t* out_ids, int mem, void* stream) {
    if (n_lists < 1 || n_queries < 0 || k_in < 1 || k < 1) return fail(RL_ERR_INVALID, "rl_merge_topk: bad sizes");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "rl_merge_topk: k must be <= 2048");
    if (n_queries == 0) return RL_OK;
    if (!in_scores || !in_ids || !out_scores || !out_ids) return fail(RL_ERR_INVALID, "rl_merge_topk: null argument");
    hipStream_t s = as_stream(stream);
    const size_t n_in = (size_t)n_lists * n_queries * k_in, n_out = (size_t)n_queries * k;
    DevBuf t_is, t_ii, t_os, t_oi;
    const float* d_is; const int32_t* d_ii; float* d_os; int32_t* d_oi;
    RL_TRY(stage_in(in_scores, n_in, mem, s, t_is, &d_is));
    RL_TRY(stage_in(in_ids, n_in, mem, s, t_ii, &d_ii));
    RL_TRY(stage_out_begin(out_scores, n_out, mem, t_os, &d_os));
    RL_TRY(stage_out_begin(out_ids, n_out, mem, t_oi, &d_oi));
    RL_TRY(launch_merge_topk(d_is, d_ii, n_lists, n_queries, k_in, k, d_os, d_oi, s));
    RL_TRY(stage_out_end(out_scores, n_out, mem, s, t_os));
    RL_TRY(stage_out_end(out_ids, n_out, mem, s, t_oi));
    return finish(mem, s);
}

int rl_topk(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int32_t k, float* out_scores,
            int32_t* out_ids, int mem, void* stream) {
    if (n_queries < 0 || n < 0 || ld < n || k < 1) return fail(RL_ERR_INVALID, "rl_topk: bad sizes");
    if (k > K_MAX) return fail(RL_ERR_UNSUPPORTED, "rl_topk: k must be <= 2048");
    if (n_queries == 0) return RL_OK;
    if ((n > 0 && !scores) || !out_scores || !out_ids) return fail(RL_ERR_INVALID, "rl_topk: null argument");
    hipStream_t s = as_stream(stream);
    const size_t n_in = (size_t)n_queries * ld, n_out = (size_t)n_queries * k;
    DevBuf t_in, t_os, t_oi;
    const float* d_in; float* d_os; int32_t* d_oi;
    RL_TRY(stage_in(scores, n_in, mem, s, t_in, &d_in));
    RL_TRY(stage_out_begin(out_scores, n_out, mem, t_os, &d_os));
    RL_TRY(stage_out_begin(out_ids, n_out, mem, t_oi, &d_oi));
    SelectWorkspace ws;  // one-off workspace: this entry point is for tests / external scorers
    {
        std::lock_guard<std::mutex> lock(g_default_opts_mu);
        ws.block_route = (int)g_default_opts.v[RL_OPT_TOPK_BLOCK];
    }
    int st = launch_topk(d_in, n_queries, n, ld, k, ws, d_os, d_oi, s);
    if (st == RL_OK) st = stage_out_end(out_scores, n_out, mem, s, t_os);
    if (st == RL_OK) st = stage_out_end(out_ids, n_out, mem, s, t_oi);
    const int sy = sync_and_drain(s);  // the workspace is freed below; bounced results reach the caller here
    select_workspace_free(ws);
    return st == RL_OK ? sy : st;
}

// ---- timing hook for bench.py ------------------------------------------------------------------------------
int rl_partition_similarity(const float* X, int64_t n, int32_t dim, const int64_t* doc_offsets, int64_t n_docs,
                            const uint8_t* nonoutlying, float* out, int mem, void* stream) {
    if (n < 0 || dim <= 0) return fail(RL_ERR_INVALID, "rl_partition_similarity: bad shape");
    if (n == 0) return RL_OK;
    if (!X || !out) return fail(RL_ERR_INVALID, "rl_partition_similarity: null argument");
    if (dim > 4096) return fail(RL_ERR_UNSUPPORTED, "rl_partition_similarity: dim must be <= 4096");
    if (!doc_offsets) n_docs = 1;
    if (n_docs < 1) return fail(RL_ERR_INVALID, "rl_partition_similarity: need at least one document");
    hipStream_t s = as_stream(stream);
    DevBuf t_x, t_off, t_sel, t_out, t_scr, t_one;
    const float* d_x; const int64_t* d_off; const uint8_t* d_sel = nullptr; float* d_out;
    RL_TRY(stage_in(X, (size_t)n * dim, mem, s, t_x, &d_x));
    if (doc_offsets) {
        RL_TRY(stage_in(doc_offsets, (size_t)n_docs + 1, mem, s, t_off, &d_off));
    } else {
        const int64_t one[2] = {0, n};
        RL_TRY(t_one.alloc(sizeof(one)));
        RL_HIP(hipMemcpyAsync(t_one.p, one, sizeof(one), hipMemcpyHostToDevice, s));
        RL_HIP(hipStreamSynchronize(s));  // `one` lives on this stack frame
        d_off = t_one.as<int64_t>();
    }
    if (nonoutlying) RL_TRY(stage_in(nonoutlying, (size_t)n, mem, s, t_sel, &d_sel));
    RL_TRY(stage_out_begin(out, (size_t)n, mem, t_out, &d_out));
    RL_TRY(t_scr.alloc(partition_sim_scratch_bytes(n, n_docs, dim)));
    RL_TRY(launch_partition_similarity(d_x, n, dim, d_off, n_docs, d_sel, d_out, t_scr.p, s));
    RL_TRY(stage_out_end(out, (size_t)n, mem, s, t_out));
    return sync_and_drain(s);  // the scratch dies with this frame
}

int rl_chunk_best_rows(rl_index* idx, const float* queries, int32_t B, const int32_t* candidates, int32_t n_cand,
                       int32_t* out_rows, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_chunk_best_rows: null index");
    if (B < 0 || n_cand < 0) return fail(RL_ERR_INVALID, "rl_chunk_best_rows: bad sizes");
    if (B == 0 || n_cand == 0) return RL_OK;
    if (!queries || !candidates || !out_rows) return fail(RL_ERR_INVALID, "rl_chunk_best_rows: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_q, t_c, t_o;
    const float* d_q; const int32_t* d_c; int32_t* d_o;
    RL_TRY(stage_in(queries, (size_t)B * idx->dim, mem, s, t_q, &d_q));
    RL_TRY(stage_in(candidates, (size_t)B * n_cand, mem, s, t_c, &d_c));
    RL_TRY(stage_out_begin(out_rows, (size_t)B * n_cand, mem, t_o, &d_o));
    RL_TRY(launch_chunk_best_rows(idx->E16 ? (const void*)idx->E16 : (const void*)idx->E, idx->E16 != nullptr, idx->dim,
                                  d_q, idx->offsets, idx->n_chunks, d_c, n_cand, (int64_t)B * n_cand, d_o, s));
    RL_TRY(stage_out_end(out_rows, (size_t)B * n_cand, mem, s, t_o));
    return finish(mem, s);
}

int rl_gather_rows(rl_index* idx, const int32_t* rows, int64_t n, float* out, int mem, void* stream) {
    if (!idx) return fail(RL_ERR_INVALID, "rl_gather_rows: null index");
    if (n < 0) return fail(RL_ERR_INVALID, "rl_gather_rows: negative count");
    if (n == 0) return RL_OK;
    if (!rows || !out) return fail(RL_ERR_INVALID, "rl_gather_rows: null argument");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    RL_TRY(use_scratch(idx, s));
    DevBuf t_r, t_o;
    const int32_t* d_r; float* d_o;
    RL_TRY(stage_in(rows, (size_t)n, mem, s, t_r, &d_r));
    RL_TRY(stage_out_begin(out, (size_t)n * idx->dim, mem, t_o, &d_o));
    RL_TRY(launch_gather_rows(idx->E16 ? (const void*)idx->E16 : (const void*)idx->E, idx->E16 != nullptr, idx->dim,
                              idx->n_rows, d_r, n, d_o, s));
    RL_TRY(stage_out_end(out, (size_t)n * idx->dim, mem, s, t_o));
    return finish(mem, s);
}

int rl_time_kernel(rl_index* idx, int kind, const float* q_dev, int32_t nq, int32_t iters, float* out_ms_total,
                   void* stream) {
    if (!idx || !q_dev || !out_ms_total || iters < 1 || nq < 1) return fail(RL_ERR_INVALID, "rl_time_kernel: bad arguments");
    hipStream_t s = as_stream(stream);
    std::lock_guard<std::mutex> lock(idx->mu);
    const bool replay_valid = idx->replay.valid;  // (use_scratch forgets the record; kind 8 touches nothing it points at)
    RL_TRY(use_scratch(idx, s));
    if (kind == 8) idx->replay.valid = replay_valid;
    const int64_t ld = (idx->n_rows + 3) & ~int64_t(3);
    const int64_t ldc = std::max<int64_t>((idx->n_chunks + 3) & ~int64_t(3), 4);
    if (kind == 0) RL_TRY(idx->scores.reserve(std::max<size_t>((size_t)idx->n_chunks * sizeof(float), 16)));
    else if (kind == 2) RL_TRY(idx->scores.reserve((size_t)2 * ldc * sizeof(float)));
    else if (kind == 3 || kind == 5 || kind == 6) RL_TRY(idx->scores.reserve((size_t)GEMM_PASS_QUERIES * ldc * sizeof(float)));
    // kind 7: sixteen queries of nq / 16 vectors each -- or, nq > 16 * 32, nq / 32 queries of 32 vectors: all their passes in ONE launch, as
    // the batch pipeline launches them
    const int32_t pp_vec = nq / PP_PASS_QUERIES <= 32 ? nq / PP_PASS_QUERIES : 32;
    const int32_t pp_n = nq / PP_PASS_QUERIES <= 32 ? PP_PASS_QUERIES : nq / 32;
    if (kind == 7) RL_TRY(idx->scores.reserve((size_t)pp_n * ldc * sizeof(float)));
    else if (kind == 0 || kind == 2 || kind == 3 || kind == 5 || kind == 6) {}
    else if (kind == 9) RL_TRY(idx->scores.reserve((size_t)(idx->n_cu > 0 ? idx->n_cu : 256) * 512 * sizeof(float)));
    else if (kind == 8) {  // replays the candidate pass of the last fused-HI row search (same queries, same thresholds)
        const auto& r = idx->replay;
        if (!r.valid || r.pools[0] != idx->misc.p || r.pools[1] != idx->fused.p || r.pools[2] != idx->pp_work.p || !hi_image_valid(idx))
            return fail(RL_ERR_UNSUPPORTED, "rl_time_kernel kind 8: no fused-HI row search ran on this index since its scratch was last resized");
    }
    else RL_TRY(idx->scores.reserve(std::max<size_t>((size_t)nq * ld * sizeof(float), 16)));
    hipEvent_t e0, e1;
    RL_HIP(hipEventCreate(&e0));
    RL_HIP(hipEventCreate(&e1));
    int st = RL_OK;
    if (kind == 2) {  // the query fragments are prepared once per batch, outside the pass
        st = pairs_prepare(idx, q_dev, nq / 2, (int64_t)(nq / 2) * idx->dim, 2, s);
        if (st != RL_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return st == RL_ERR_UNSUPPORTED ? fail(st, "rl_time_kernel: the pair kernel does not apply to this index / shape") : st; }
    }
    if (kind == 3 || kind == 5 || kind == 6) {  // eight queries of nq / 8 vectors each
        st = gemm_prepare(idx, q_dev, nq / GEMM_PASS_QUERIES, (int64_t)(nq / GEMM_PASS_QUERIES) * idx->dim, GEMM_PASS_QUERIES, s, true);
        if (st == RL_OK && !image_valid(idx)) st = RL_ERR_UNSUPPORTED;  // (an index of rows + HI image: only the sixteen-query pass applies)
        if (st != RL_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return st == RL_ERR_UNSUPPORTED ? fail(st, "rl_time_kernel: the eight-query kernel does not apply to this index / shape") : st; }
    }
    if (kind == 7) {
        st = pp_vec >= 1 && pp_vec * pp_n == nq ? gemm_prepare(idx, q_dev, pp_vec, (int64_t)pp_vec * idx->dim, pp_n, s) : RL_ERR_UNSUPPORTED;
        if (st == RL_OK && !(approx_image_valid(idx) && idx->dim >= 256)) st = RL_ERR_UNSUPPORTED;
        if (st != RL_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return st == RL_ERR_UNSUPPORTED ? fail(st, "rl_time_kernel: the sixteen-query kernel does not apply to this index / shape") : st; }
    }
    if (kind == 4 || kind == 10) {  // (the lazy image is built BEFORE the timed region, like gemm_prepare's above)
        st = demand_images(idx, IMG_HI_PLANE, s);
        if (st != RL_OK) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return st; }
    }
    RL_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters && st == RL_OK; ++i) {
        if (kind == 7) st = launch_maxsim_pp(approx_image(idx), idx->n_rows, idx->dim, idx->qplanes.p, pp_n, 0, pp_n,
                                             pp_vec, idx->row_to_chunk, idx->offsets, idx->ends.as<uint32_t>(), idx->scores.as<float>(), ldc,
                                             idx->n_cu, s, approx_scale(idx));
        else if (kind == 3) st = gemm_pass(idx, nq / GEMM_PASS_QUERIES, GEMM_PASS_QUERIES, 0, GEMM_PASS_QUERIES, idx->scores.as<float>(), ldc, s);
        else if (kind == 8) {
            const auto& r = idx->replay;
            if (hipMemsetAsync(r.cnt, 0, (size_t)r.B * sizeof(uint32_t), s) != hipSuccess) st = RL_ERR_HIP;  // the lists fill up again on every run
            else if (r.pp && r.round1_tiles > 0) {  // both rounds with the thresholds each of them ran with (the ranking between them is not replayed)
                CandArgs ca1 = r.ca;
                ca1.tau = r.thr1;
                st = launch_pp_rows_pass(idx->hi_image.p, idx->n_rows, idx->dim, r.B, r.qs, idx->norm, r.mode, &ca1, idx->pp_work.p, r.log_cap, idx->n_cu, s,
                                         idx->split_scale, 0, r.round1_tiles, false, r.row_test);
                if (st == RL_OK)
                    st = launch_pp_rows_pass(idx->hi_image.p, idx->n_rows, idx->dim, r.B, r.qs, idx->norm, r.mode, &r.ca, idx->pp_work.p, r.log_cap, idx->n_cu, s,
                                             idx->split_scale, r.round1_tiles, -1, true, r.row_test);
            }
            else if (r.pp) st = launch_pp_rows_pass(idx->hi_image.p, idx->n_rows, idx->dim, r.B, r.qs, idx->norm, r.mode, &r.ca, idx->pp_work.p, r.log_cap,
                                                    idx->n_cu, s, idx->split_scale, 0, -1, false, r.row_test);
            else st = launch_score_planes_pass(idx->hi_image.p, idx->n_rows, idx->dim, r.B, r.qs, nullptr, 0, idx->norm, idx->sumsq, r.mode, 1, nullptr, &r.ca,
                                               idx->n_cu, s, idx->split_scale, true, true);
        }
        else if (kind == 0) st = maxsim_scores_device(idx, q_dev, nq, idx->scores.as<float>(), s);
        else if (kind == 9) st = launch_mfma_f16_rate(idx->scores.as<float>(), idx->n_cu, MFMA_RATE_ITERS, s, nullptr);
        else if (kind == 2) st = pairs_pass(idx, nq / 2, 2, 0, idx->scores.as<float>(), ldc, s);
        else if (kind == 5 || kind == 6) {  // the approximate MaxSim pass of a batch: eight queries over the HI image (5: two MFMA products, 6: one)
            st = hi_image_valid(idx) ? launch_maxsim_gemm(idx->hi_image.p, idx->n_rows, idx->dim, idx->qplanes.p, GEMM_PASS_QUERIES, 0,
                                                          GEMM_PASS_QUERIES, nq / GEMM_PASS_QUERIES, idx->row_to_chunk, idx->offsets,
                                                          idx->ends.as<uint32_t>(), idx->scores.as<float>(), ldc, idx->n_cu, s, idx->split_scale, true,
                                                          nullptr, kind == 6)
                                     : fail(RL_ERR_UNSUPPORTED, "rl_time_kernel: the index has no HI image");
        }
        else if (kind == 4) {  // the ranking pass of the half-bytes search: the f16 stream kernel over the HI plane
            st = hi_valid(idx) ? launch_maxsim_stream16(idx->hiplane.as<uint16_t>(), idx->n_rows, idx->dim, q_dev, nq, idx->row_to_chunk, idx->offsets,
                                                        idx->n_chunks, 1, idx->scores.as<float>(), ld, idx->n_cu, s)
                               : fail(RL_ERR_UNSUPPORTED, "rl_time_kernel: the index has no HI plane");
        }
        else if (kind == 10) {  // the approximate pass of the few-queries MaxSim route: ONE query of nq vectors, MaxSim over the HI plane
            st = hi_valid(idx) && nq <= 32 ? launch_maxsim_stream_batch(idx->hiplane.p, true, idx->n_rows, idx->dim, q_dev, nq, (int64_t)nq * idx->dim, 1,
                                                                        idx->row_to_chunk, idx->offsets, idx->n_chunks, idx->scores.as<float>(), ldc, idx->n_cu, s,
                                                                        0.f, nullptr)
                                           : fail(RL_ERR_UNSUPPORTED, "rl_time_kernel: the index has no HI plane (or nq > 32)");
        }
        else st = score_rows(idx, q_dev, nq, ld, s);
    }
    RL_HIP(hipEventRecord(e1, s));
    RL_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    RL_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *out_ms_total = ms;
    return st;
}

}  // extern "C"
