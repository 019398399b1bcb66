/*
 * raglite_hip.h -- C ABI of libraglite_hip.so: the MI355X (gfx950) retrieval / rerank hot path
 * of RAGLite.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (superlinear-ai/raglite v1.0.0) is pure Python and has no FFI of its own; each
 * entry point below states the reference lines (relative to /root/reference) whose arithmetic
 * it replaces.  INTEGRATION.md shows the ctypes binding a RAGLite maintainer would add.
 *
 * Conventions
 *  - every function returns 0 (RL_OK) or a negative rl_status; rl_last_error() gives the
 *    thread-local message of the last failure on the calling thread.
 *  - `mem` says where the caller's data pointers live: RL_MEM_HOST (the library stages through
 *    its own device scratch, synchronously) or RL_MEM_DEVICE (pointers are HIP device pointers,
 *    work is enqueued on `stream` and NOT synchronised -- the caller owns ordering).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - all matrices are row-major and dense; dim is the embedding dimension.
 *  - thread safety: calls on distinct rl_index handles / distinct streams may run concurrently;
 *    calls on one handle share its device scratch: they serialise on an internal mutex, and a call on
 *    another stream than the handle's previous call first waits for that stream (the reference calls the path from up to
 *    4 worker threads: src/raglite/_insert.py:159,208-237; src/raglite/_rag.py:317-318).
 */
#ifndef RAGLITE_HIP_H
#define RAGLITE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RL_OK = 0,
    RL_ERR_INVALID = -1,     /* bad argument (maps to Python ValueError) */
    RL_ERR_HIP = -2,         /* HIP runtime failure (RuntimeError) */
    RL_ERR_UNSUPPORTED = -3, /* shape outside what the kernels implement (ValueError) */
    RL_ERR_NOMEM = -4        /* device allocation failed (MemoryError) */
} rl_status;

typedef enum { RL_MEM_HOST = 0, RL_MEM_DEVICE = 1 } rl_mem;
/* How an index multiplies queries with its corpus in the MFMA streaming kernel (rl_index_set_arithmetic). */
typedef enum {
    RL_ARITH_AUTO = 0,       /* request: RL_ARITH_F16_SPLIT where the corpus allows it, else RL_ARITH_FP32_EXACT */
    RL_ARITH_FP32_EXACT = 1, /* v_mfma_f32_16x16x4_f32: bitwise an ordered fp32 fmaf chain */
    RL_ARITH_F16_SPLIT = 2,  /* in effect only: fp32 operands as fp16 (hi, lo) pairs, see below */
    RL_ARITH_F16_STORED = 3  /* in effect only: an rl_index_create_f16 index */
} rl_arith;
typedef enum { RL_F32 = 0, RL_F16 = 1 } rl_dtype;
/* src/raglite/_config.py:69 `vector_search_distance_metric`; src/raglite/_typing.py:123-134 */
typedef enum { RL_COSINE = 0, RL_DOT = 1, RL_L2 = 2 } rl_metric;
typedef enum { RL_SYNTH_UNIFORM = 0, RL_SYNTH_SMALL_INT = 1 } rl_synth_kind;

typedef struct rl_index rl_index; /* opaque device-resident index (corpus matrix + chunk CSR) */

/* ---- runtime ------------------------------------------------------------------------------ */
int rl_version(void);
const char* rl_last_error(void);
/* Select the HIP device for the calling thread; fails if it is not a gfx950 part. */
int rl_init(int device);
int rl_device_count(int* count);
/* name[<=len], number of CUs, bytes of device memory */
int rl_device_info(int device, char* name, int len, int* compute_units, int64_t* total_mem);

/* Device memory helpers so that a pure-ctypes caller needs no other GPU library. */
int rl_dev_alloc(void** ptr, size_t bytes);
int rl_dev_free(void* ptr);
int rl_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream);
int rl_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int rl_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int rl_stream_sync(void* stream);

/* Counter-based synthetic data, bit-identical to oracle/oracle.py:synth_uniform / synth_small_int:
 * element i (i in [start, start+count)) of stream `seed`.  dst is a DEVICE pointer. */
int rl_synth_fill(float* dst, int64_t start, int64_t count, uint64_t seed, int kind, void* stream);

/* ---- a1 + a2 + a3: late-chunking pool, L2 normalise, fp16 cast ------------------------------
 * Replaces src/raglite/_embed.py:131-140 (per-sentence np.mean over contiguous token rows,
 * rowwise X /= ||X||, astype(float16)) and :154,158-164 (whole-string mean, eps-guarded
 * normalise).  The host computes the row spans (largest-remainder split, :122-129) and passes
 * them in; span s pools token rows [span_begin[s], span_end[s]).
 *   tokens      [T x dim] f32
 *   normalize   0/1            (RAGLiteConfig.embedder_normalize)
 *   eps         0  -> unguarded divide (:139);  >0 -> divide by max(norm, eps) (:160-163)
 *   out_f32     [S x dim] f32 or NULL;  out_f16 [S x dim] IEEE half bits or NULL
 * Accumulation is fp64 (the reference pools float64 arrays), the cast rounds to nearest even.
 * An empty span produces NaN, as np.mean of zero rows does. */
int rl_pool_norm(const float* tokens, int64_t n_token_rows, int32_t dim,
                 const int64_t* span_begin, const int64_t* span_end, int64_t n_spans,
                 int32_t normalize, double eps, float* out_f32, uint16_t* out_f16,
                 int mem, void* stream);

/* ---- a5: query adapter -----------------------------------------------------------------------
 * Replaces src/raglite/_search.py:62  `(Q @ q).astype(q.dtype)`; out[b] = A @ q[b].
 *   A [dim x dim] f32, queries [B x dim] f32, out_f32 [B x dim] or NULL, out_f16 or NULL. */
int rl_adapter_apply(const float* A, const float* queries, int32_t n_queries, int32_t dim,
                     float* out_f32, uint16_t* out_f16, int mem, void* stream);

/* ---- index lifecycle -------------------------------------------------------------------------
 * The device-resident equivalent of the `chunk_embedding` table (src/raglite/_database.py:403-430):
 * one row per chunklet vector, rows of a chunk contiguous (a4: src/raglite/_split_chunks.py:116-122,
 * src/raglite/_insert.py:114-123).
 *   embeddings     [n_rows x dim] f32; with mem == RL_MEM_DEVICE the pointer is BORROWED (must
 *                  outlive the index), with RL_MEM_HOST it is copied to the device.
 *   chunk_offsets  [n_chunks + 1] int64 ascending CSR, chunk_offsets[0]==0, [n_chunks]==n_rows
 *                  (always a HOST pointer); NULL -> every row is its own chunk.
 *   metric         rl_metric used by rl_search_rows / rl_search_chunks.
 * Precomputes ||e|| per row (cosine) or ||e||^2 (l2): 4 B/row extra. */
int rl_index_create(rl_index** out, const float* embeddings, int64_t n_rows, int32_t dim,
                    const int64_t* chunk_offsets, int64_t n_chunks, int metric, int mem, void* stream);
/* fp16-STORED index.  The reference stores its embeddings as float16
 * (src/raglite/_embed.py:140, src/raglite/_database.py:279-283), so this storage is lossless for real
 * RAGLite data and halves the bytes of every HBM-bound pass.  embeddings_f16: [n_rows x dim] IEEE
 * binary16 bit patterns; dim must be one of 128, 256, 384, 512, 768, 1024 or (round 6: 1536- / 3072-wide embedders) a multiple of 128 up
 * to 4096.  Queries, scores and every other argument stay fp32; arithmetic is fp32 (VALU) or fp16 x fp16 -> fp32 MFMA (exact products).
 * rl_index_append on such an index takes fp32 rows and rounds them to nearest-even fp16;
 * rl_maxsim_rerank needs nq <= 32 and dim % 16 == 0 on it (dim 128: the cfg 3 kernel); beyond dim 1024 every MaxSim call takes up to 32
 * query vectors. */
int rl_index_create_f16(rl_index** out, const uint16_t* embeddings_f16, int64_t n_rows, int32_t dim,
                        const int64_t* chunk_offsets, int64_t n_chunks, int metric, int mem, void* stream);
int rl_index_destroy(rl_index* index);
int rl_index_info(const rl_index* index, int64_t* n_rows, int32_t* dim, int64_t* n_chunks, int* metric);
/* Device memory of an index, in bytes: out[0] the stored rows (borrowed or owned), out[1] the pre-split corpus image (0: not built),
 * out[2] the image of the hi halves, out[3] the row-major HI plane, out[4] the per-call scratch grown so far, out[5] / out[6] free /
 * total device memory now, out[7] the headroom an image has to leave free to be built.  The three images are optional accelerators
 * (4 + 2 + 2 bytes per element next to the 4 of an fp32 corpus): each is built only while it leaves max(2 GiB, 1/16 of the device) --
 * RL_OPT_IMAGE_HEADROOM_MB overrides -- free for the scratch and the caller; without them the same calls run through the kernels over
 * the stored rows (same results: every image path is bit-identical to, or re-scored exactly against, the rows).
 * Index footprint: all three images make an fp32 index 3 x its corpus.  The headline path -- MaxSim batches -- needs only the HI image:
 * with RL_OPT_KEEP_IMAGE = 0 and RL_OPT_KEEP_HI_PLANE = 0 an fp32 index of dim 256 / 384 / 512 / 768 / 1024 -- or a WIDE one, see below --
 * holds rows + HI image (1.5 x) and rl_maxsim_topk_batch / rl_maxsim_batch_begin run the same pipeline with the same results: approximate
 * pass over the HI image, exact re-scoring over the rows; only the guarded full-precision fallback (list overflow, unusable bound) changes
 * kernel -- the streaming kernels over the rows (a wide index: the exact re-scoring kernel over every chunk) instead of the eight-query
 * pass over the pre-split image.  Row searches on such an index take the kernels over the rows (B <= 16: the full fp32 pass; B >= 96:
 * score_gemm instead of the fused top-k).
 * WIDE indexes (round 6): the half-bytes routes -- HI image / HI plane, bound-filtered MaxSim batches, few-queries and fused row searches --
 * take any dim % 32 == 0 up to 1024 and, beyond, dim % 128 == 0 up to 4096 (1536- / 3072-wide embedders, src/raglite/_embed.py:155-158);
 * on a wide index even one MaxSim query goes through the sixteen-query pass and the few-queries row search takes up to 4 queries (16 at
 * dim <= 1024).  Other widths and k > 512 run on the full-precision routes: same results, 2-3 x slower. */
int rl_index_memory(const rl_index* index, int64_t out[8]);

/* Warm-up for lazy images (RL_OPT_LAZY_IMAGES, the default): build the images in `images` (RL_IMAGE_* bits) NOW, on `stream`, instead of
 * inside the first search whose route reads them -- that first call otherwise allocates device memory (0.5 x / 1 x the corpus per image),
 * makes one pass over the rows and synchronises the stream once (the norm statistics of the error bounds come back to the host), inside
 * an entry point that is asynchronous from then on.  `built` (nullable) receives the bits of the images the index holds afterwards: an
 * image the options / the shape / the free memory do not allow is simply absent (never an error -- the routes over the stored rows
 * answer, same results).  An image skipped for lack of room (it would not have left RL_OPT_IMAGE_HEADROOM_MB free) is asked for again
 * by later searches -- at most one call in 64 retries -- and by every rl_index_prepare.  Which images a workload reads: MaxSim batches
 * RL_IMAGE_HI; row searches of <= 16 queries RL_IMAGE_HI_PLANE, of >= 96 queries RL_IMAGE_PRESPLIT | RL_IMAGE_HI (table above). */
#define RL_IMAGE_PRESPLIT 1u
#define RL_IMAGE_HI 2u
#define RL_IMAGE_HI_PLANE 4u
int rl_index_prepare(rl_index* index, uint32_t images, uint32_t* built, void* stream);

/* ---- index lifecycle beyond create/destroy -----------------------------
 * rl_index_append: the device image of `insert_documents` appending `chunk_embedding` rows
 * (src/raglite/_insert.py:247-272, rows ordered by chunk, src/raglite/_database.py:403-430).
 *   rows             [n_new_rows x dim] f32 (host or device per `mem`)
 *   new_chunk_sizes  HOST int64[n_new_chunks], sum == n_new_rows; NULL -> one chunk per new row.
 * New rows / chunks get the next ordinals; existing ordinals never change.  The index takes
 * ownership of its storage on the first append (a borrowed device matrix is copied once) and grows
 * geometrically afterwards.
 * rl_index_delete_chunks: the device image of `delete_documents` (src/raglite/_delete.py:148-176).
 * The chunks' rows stay in place as tombstones (ordinals stay stable) and never match again in any
 * search of this index; deleting twice is a no-op.  chunk_ordinals is a HOST array.
 * rl_index_live: rows / chunks that are not tombstoned. */
int rl_index_append(rl_index* index, const float* rows, int64_t n_new_rows, const int64_t* new_chunk_sizes,
                    int64_t n_new_chunks, int mem, void* stream);
int rl_index_delete_chunks(rl_index* index, const int64_t* chunk_ordinals, int64_t n, void* stream);
int rl_index_live(rl_index* index, int64_t* live_rows, int64_t* live_chunks, void* stream);
/* rl_index_compact: reclaim the tombstones.  The surviving chunks keep their relative order and get the ordinals
 * 0 .. live_chunks - 1, their rows move together (one gather pass over the matrix, the per-row norms travel with
 * them), every derived structure is rebuilt; afterwards the index equals one created from the surviving rows.
 *   out_remap   HOST int64[old n_chunks] or NULL: new ordinal of every old chunk, -1 for a deleted one
 * A long-lived index under delete_documents traffic (src/raglite/_delete.py:148-176) would otherwise keep streaming
 * its dead rows in every scan.  No-op (identity remap) without tombstones.  Synchronous. */
int rl_index_compact(rl_index* index, int64_t* out_remap, int64_t* new_n_rows, int64_t* new_n_chunks, void* stream);

/* ---- arithmetic of the MFMA streaming kernel over an fp32-stored corpus --------------------------
 * The reference multiplies in fp32 (DuckDB FLOAT[d], src/raglite/_typing.py:99-134) or fp64 (NumPy,
 * src/raglite/_query_adapter.py:174); BASELINE.json asks for scores within 1e-4.  Two ways to get there:
 *   RL_ARITH_FP32_EXACT  fp32 MFMAs, bitwise an ordered fmaf chain; matrix-pipe-bound at 32 query vectors.
 *   RL_ARITH_F16_SPLIT   every fp32 operand x is scaled by a power of two and written as hi + lo with hi, lo
 *                        fp16 (22 significant bits); e.q = eh.qh + (el.qh + eh.ql) on the fp16 matrix pipe,
 *                        whose fp16 x fp16 products are exact and accumulate in fp32.  Measured error against
 *                        float64 is that of the fp32 chain; the kernel becomes
 *                        HBM-bound (10 % faster at 32 x 1M x 1024).
 * Default RL_ARITH_AUTO: F16_SPLIT when every element is finite and the largest elements of all non-zero rows
 * lie within a factor 2^10 of each other (any normalised corpus), FP32_EXACT otherwise;
 * rl_set_default_option(RL_OPT_ARITHMETIC, RL_ARITH_FP32_EXACT) makes FP32_EXACT the start value of every index created
 * afterwards.  The batched GEMM (>= 96 queries) follows the
 * same setting (22 -> 11 ms at 1000 x 1.25 M x 1024) and so does rl_maxsim_rerank's dim-128 MFMA kernel (718 k -> 875 k
 * queries/s at 32 x (256 x 64) x 128); the single-query VALU scan always computes in fp32.
 * rl_index_set_arithmetic: mode = RL_ARITH_AUTO | RL_ARITH_FP32_EXACT.
 * rl_index_arithmetic: what is in effect (FP32_EXACT, F16_SPLIT or F16_STORED). */
int rl_index_set_arithmetic(rl_index* index, int mode);
int rl_index_arithmetic(rl_index* index, int* in_effect);

/* ---- options ---------------------------------------------------------------------------------------------------------------
 * Every search below has ONE result whatever route it takes (exact top-k of exactly computed scores); the routes differ in speed
 * and memory.  The options choose routes: they exist for A/B measurements, for tests that must force a fallback, and for deployments
 * that want a smaller index.  NO environment variable is read by the library: an option is a property of an index, fixed when the
 * index is created from the process-wide defaults (rl_set_default_option: affects indexes created AFTERWARDS) and changed only
 * through rl_index_set_option -- which takes the index' mutex like every call on the handle, so a change never lands in the middle
 * of a search.  Timing-experiment builds of the kernels (instantiations that skip work and return WRONG results) are not in this
 * library at all: they are compiled only with -DRAGLITE_EXPERIMENTS (raglite_amd._build.build(experiments=True) ->
 * libraglite_hip_exp.so, used by scripts/gpu_calls/ only).
 *   key                         values (default)   what it routes
 *   RL_OPT_HI_SEARCH            0 / 1 (1)          B <= 16 row searches rank on the fp16 HI plane + exact re-scoring (else: full pass)
 *   RL_OPT_HI_MAXSIM            0 / 1 (1)          MaxSim batches rank on the HI image + exact re-scoring (else: full-precision passes)
 *   RL_OPT_HI_FEW               0 / 1 (1)          ... and ONE or TWO MaxSim queries (rl_maxsim_topk, batches of < 3) rank from the row-major HI plane
 *                                                  (2 B per element, HBM-bound) and re-score exactly (0: the streaming kernels over the fp32 rows)
 *   RL_OPT_TOPK_BLOCK           0 / 1 / 2 (2)      exact top-k of <= 262 144 scores per query (MaxSim chunk scores, the fused top-k's sample, rl_topk) in
 *                                                  ONE launch, one block per query (0: histogram / filter / final, three launches; same results);
 *                                                  2: the block first cuts the scores to the ~k that reach the k-th largest of its 1024 thread maxima
 *   RL_OPT_HI_PIVOT             0 / 1 (1)          the B <= 16 row search takes its candidate threshold from the k-th largest of ~500 workgroup maxima of the
 *                                                  approximate similarities (two launches, k <= 128) instead of ranking them exactly first (three); same results
 *   RL_OPT_HI_PRODUCTS          1 / 2 (1)          fp16 MFMA products per multiply in that approximate pass
 *   RL_OPT_PP_PASS              0 / 1 (1)          its sixteen-query kernel (maxsim_pp.hip; 0: the eight-query kernel)
 *   RL_OPT_FUSED_TOPK           0 / 1 (1)          B >= 96 row searches keep candidate lists instead of a score matrix
 *   RL_OPT_FUSED_HI             0 / 1 (1)          ... with both GEMM passes over the HI image at one product
 *   RL_OPT_FUSED_PP             0 / 1 (1)          ... and the candidate pass on the sixteen-group tile of maxsim_pp.hip
 *   RL_OPT_FUSED_PP_SAMPLE      0 / 1 (1)          ... and the sample pass before it on the same tile (0: the eight-group kernel of maxsim_gemm.hip)
 *   RL_OPT_LIST_SELECT          0 / 1 (1)          ... whose per-query lists are cut by a radix SELECT of their k-th best score (0: by sorting them)
 *   RL_OPT_FUSED_TWO_ROUNDS     0 / 1 (1)          ... in two rounds: thresholds tightened from the first 3/16 of the rows (0: one round)
 *   RL_OPT_FUSED_TOPK_CAP       0 | 1..8192 (0)    list capacity of the fused top-k (0: built-in; tests force overflows with it)
 *   RL_OPT_FUSED_TOPK_STRIDE    0 | >= 2 (0)       sample stride of the fused top-k (0: built-in rule)
 *   RL_OPT_GEMM_PASS            0 / 1 (1)          MaxSim batches of >= 3 queries share passes over the pre-split image
 *   RL_OPT_QUERY_PAIRS          0 / 1 (1)          two-query passes of the streaming kernel
 *   RL_OPT_PLANES_GEMM          0 / 1 (1)          dense row-score GEMM over the image for B >= 96 (else: score_gemm over the rows)
 *   RL_OPT_KEEP_IMAGE           0 / 1 (1)          keep the pre-split corpus image (4 B per element; 0 releases it -- MaxSim batches of an
 *                                                  fp32 index then run on rows + HI image alone, see "index footprint" below)
 *   RL_OPT_KEEP_HI              0 / 1 (1)          keep the HI plane and the HI image (2 + 2 B per element; 0 releases them)
 *   RL_OPT_KEEP_HI_PLANE        0 / 1 (1)          keep the row-major HI plane (2 B per element; what B <= 16 row searches rank on)
 *   RL_OPT_IMAGE_HEADROOM_MB    -1 | >= 0 (-1)     device memory the images must leave free (-1: max(2 GiB, 1/16 of the device))
 *   RL_OPT_ARITHMETIC           rl_arith (AUTO)    same as rl_index_set_arithmetic
 *   RL_OPT_PAIRS_PACKED         0 / 1 / 2 (2)      exact re-scoring of (query, chunk) pairs packs the candidates' rows into shared 16-row MFMA
 *                                                  tiles (0: every chunk its own tiles; same bits); 2: sixteen waves per workgroup instead of eight
 *                                                  (fp32 rows, dim % 128 == 0)
 *   RL_OPT_EXACT_KTH_THRESHOLD  0 / 1 (1)          MaxSim batches: second, tighter candidate threshold from the EXACT scores of the
 *                                                  approximate top-k (exact k-th - m instead of approximate k-th - 2 m)
 *   RL_OPT_F16_EXACT            0 / 1 (1)          rl_maxsim_topk_batch_f16 over an fp16-stored index returns the one-product pass's own top-k
 *                                                  (exact: fp16 x fp16 products, fp32 sums); 0: the bound-filtered pipeline with exact re-scoring
 *   RL_OPT_LAZY_IMAGES          0 / 1 (1)          an image is built by the FIRST call whose route reads it (allocation + one pass over the rows,
 *                                                  ~2 ms per image at 1 M x 1024, inside that call) instead of with the index: MaxSim batches ask
 *                                                  for the HI image, row searches of <= 16 queries for the HI plane, of >= 96 queries for the
 *                                                  pre-split image + HI image -- a MaxSim-only or a single-query deployment keeps 1.5 x the corpus
 *                                                  (0: every image the KEEP_* options allow is built with the index: 3 x, no first-call cost)
 * KEEP_* and IMAGE_HEADROOM_MB rebuild / release the images at once (synchronous).  Unknown key or a value outside the column above:
 * RL_ERR_INVALID.  rl_index_get_option returns what is set (not whether a route is usable on this index: rl_index_memory and
 * rl_index_filter_stats report that). */
typedef enum {
    RL_OPT_HI_SEARCH = 1, RL_OPT_HI_MAXSIM = 2, RL_OPT_HI_PRODUCTS = 3, RL_OPT_PP_PASS = 4, RL_OPT_FUSED_TOPK = 5, RL_OPT_FUSED_HI = 6,
    RL_OPT_FUSED_PP = 7, RL_OPT_FUSED_TOPK_CAP = 8, RL_OPT_FUSED_TOPK_STRIDE = 9, RL_OPT_GEMM_PASS = 10, RL_OPT_QUERY_PAIRS = 11,
    RL_OPT_PLANES_GEMM = 12, RL_OPT_KEEP_IMAGE = 13, RL_OPT_KEEP_HI = 14, RL_OPT_IMAGE_HEADROOM_MB = 15, RL_OPT_ARITHMETIC = 16,
    RL_OPT_EXACT_KTH_THRESHOLD = 17, RL_OPT_FUSED_TWO_ROUNDS = 18, RL_OPT_KEEP_HI_PLANE = 19, RL_OPT_PAIRS_PACKED = 20,
    RL_OPT_F16_EXACT = 21, RL_OPT_LAZY_IMAGES = 22, RL_OPT_FUSED_PP_SAMPLE = 23,
    RL_OPT_LIST_SELECT = 24, RL_OPT_HI_FEW = 25, RL_OPT_TOPK_BLOCK = 26, RL_OPT_HI_PIVOT = 27, RL_OPT_COUNT_ = 28
} rl_option;
int rl_set_default_option(int key, int64_t value);
int rl_get_default_option(int key, int64_t* value);
int rl_index_set_option(rl_index* index, int key, int64_t value);
int rl_index_get_option(rl_index* index, int key, int64_t* value);

/* ---- a6 + a7: similarity + exact row top-k ----------------------------------------------------
 * Replaces the SQL at src/raglite/_search.py:69-79 (`sim = 1 - dist`, ORDER BY dist LIMIT k) with
 * dist per src/raglite/_typing.py:123-134, ranked EXACTLY (the reference's HNSW is approximate).
 *   queries [B x dim] f32;  k <= 2048
 *   out_scores [B x k] f32 (sim, descending), out_rows [B x k] int32 (row ordinals; ties ->
 *   lowest row); when k > n_rows the tail is filled with score -inf / row -1.
 * How the rows are found does not change what is returned: up to 16 queries over a big fp32 corpus
 * rank on an fp16 "HI plane" of the corpus (2 B per element, kept by the index), bound the error
 * rigorously and re-score the candidates with the exact kernels; 96 or more queries rank through a
 * GEMM with fused candidate lists instead of a score matrix.  Both are bit-identical to the plain
 * pass + selection and fall back to it on the device where their bounds do not hold. */
int rl_search_rows(rl_index* index, const float* queries, int32_t n_queries, int32_t k,
                   float* out_scores, int32_t* out_rows, int mem, void* stream);

/* ---- the order-first cut over a corpus SHARDED across several indexes (SURVEY.md section 8e with 8f-1's rank_limit) -----
 * `rank_limit` of rl_search_rows_ranked is the reference's `ORDER BY dist LIMIT 1_000_000` over the WHOLE table
 * (src/raglite/_search.py:120-141); applied per shard it would admit up to world x rank_limit rows.  The cut is a three-level radix
 * select (11 + 11 + 10 bits of the order-preserving score key) whose histograms are additive, so the shards walk to the GLOBAL
 * threshold together when the caller sums each level over them:
 *     rl_rank_cut_begin(index, queries, B)                                  similarities of every live row, kept by the index
 *     for level in 0, 1, 2:
 *         rl_rank_cut_level(index, level, rank_limit, hist)                 hist [B x 2048] uint32: this shard's histogram of the level
 *         <sum hist over the shards: rl_allreduce_sum_u32, or any all-reduce>
 *         rl_rank_cut_level_done(index, level, hist)                        the summed histogram back
 *     rl_rank_cut_ties(index, rank_limit, ties)                             ties [B]: this shard's rows ON the threshold key
 *     <all-gather ties; ties_before[q] = sum of ties[q] over the shards holding LOWER global rows>
 *     rl_rank_cut_finish(index, rank_limit, ties_before, chunk_filter, k, out_scores, out_rows)
 * out_* [B x k]: this shard's top-k among the rows that are inside the global cut (ties on the threshold taken in global row order,
 * as the single-index cut takes them) and pass the filter; merging the shards' lists (rl_allgather_merge_topk) gives bit for bit what
 * ONE index over the whole corpus returns for rl_search_rows_ranked.  rank_limit must be smaller than the total number of rows of all
 * shards (otherwise there is no cut: call rl_search_rows_filtered).  The calls of one search must not be interleaved with other
 * searches on the same index; B x n_rows x 4 bytes of scores must fit one score batch (8 GB). */
int rl_rank_cut_begin(rl_index* index, const float* queries, int32_t n_queries, int mem, void* stream);
int rl_rank_cut_level(rl_index* index, int level, int64_t rank_limit, uint32_t* out_hist, int mem, void* stream);
int rl_rank_cut_level_done(rl_index* index, int level, const uint32_t* hist_sum, int mem, void* stream);
int rl_rank_cut_ties(rl_index* index, int64_t rank_limit, uint32_t* out_ties, int mem, void* stream);
int rl_rank_cut_finish(rl_index* index, int64_t rank_limit, const uint32_t* ties_before, const uint32_t* chunk_filter, int32_t k,
                       float* out_scores, int32_t* out_rows, int mem, void* stream);

/* ---- a6 + a7 + a8: the reference's two-stage semantics ------------------------------------------
 * top-`num_hits` rows -> max(sim) GROUP BY chunk -> top-`k` chunks (src/raglite/_search.py:66-67,
 * 75-79,143-149).  out_counts[b] (<= k) chunks are valid per query; the rest is -inf / -1. */
int rl_search_chunks(rl_index* index, const float* queries, int32_t n_queries, int32_t num_hits,
                     int32_t k, float* out_scores, int32_t* out_chunks, int32_t* out_counts,
                     int mem, void* stream);

/* ---- a9: MaxSim late interaction ----------------------------------------------------------------
 * score[c] = sum_{i<nq} max_{j in chunk c} Q[i].D[j]  -- the multi-query-vector generalisation of
 * src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174, offered behind the reranker
 * plugin boundary (src/raglite/_search.py:394-396).  Dot-product similarity (ColBERT convention:
 * rows are pre-normalised), independent of the index metric.
 *
 * rl_maxsim_topk: one query (nq vectors) against EVERY chunk of the index, exact top-k chunks.
 *   out_scores [k] f32 descending, out_chunks [k] int32 (ties -> lowest chunk ordinal).
 * rl_maxsim_scores: the same scores, all chunks, no selection: out_scores [n_chunks]. */
int rl_maxsim_topk(rl_index* index, const float* query_vecs, int32_t nq, int32_t k,
                   float* out_scores, int32_t* out_chunks, int mem, void* stream);
int rl_maxsim_scores(rl_index* index, const float* query_vecs, int32_t nq, float* out_scores,
                     int mem, void* stream);
/* rl_maxsim_topk_batch: `n_queries` independent queries (each nq vectors) against every chunk, then ONE batched
 * selection launch for all of them.  In RL_ARITH_F16_SPLIT arithmetic eight queries (nq <= 32) share one pass over
 * the index' pre-split corpus image (fp16 hi | lo planes written when the index is built: 4 more bytes per element
 * of device memory; RL_OPT_KEEP_IMAGE = 0 releases it), otherwise two queries or one query
 * take a pass over the fp32 / fp16 rows.  On a big fp32 index (>= 64 M elements) the passes of a batch of three or
 * more queries run over an image of the hi halves only (2 more bytes per element; ONE fp16 MFMA product per multiply
 * instead of three -- q_hi . e_hi; RL_OPT_HI_PRODUCTS = 2: two), SIXTEEN queries per pass (maxsim_pp.hip; dim >= 256;
 * RL_OPT_PP_PASS = 0 or smaller dims: eight, maxsim_gemm.hip), every chunk's score error is bounded rigorously from
 * what the hi halves of corpus and queries drop, and the chunks that could be in the top-k are
 * re-scored with exact fp32 products: the same top-k, scores as accurate as before; where the bound does not decide
 * (thousands of near-identical chunks) the full-precision passes run instead, on the device.
 * RL_OPT_HI_MAXSIM = 0 / RL_OPT_KEEP_HI = 0 switch that off.  A big fp16-STORED index (rl_index_create_f16) takes the same
 * pipeline with its stored halves as that image: the approximate pass multiplies q_hi . e (what it drops, q_lo . e, is bounded the same
 * way), the candidates are re-scored over the stored rows, the two-product passes are the guarded fallback.
 *   query_vecs [n_queries x nq x dim] f32; out_scores / out_chunks [n_queries x k]. */
int rl_maxsim_topk_batch(rl_index* index, const float* query_vecs, int32_t n_queries, int32_t nq, int32_t k,
                         float* out_scores, int32_t* out_chunks, int mem, void* stream);
/* rl_maxsim_topk_batch_f16: the same search for queries that ARE IEEE fp16 values -- what the reference's embed_strings returns
 * (src/raglite/_embed.py:140,164: `astype(np.float16)`) and what its query adapter hands on (src/raglite/_search.py:62 casts the adapted
 * query back to the query's dtype).  The queries are widened on the device (exact) and take the route of rl_maxsim_topk_batch -- with one
 * difference over an fp16-STORED index (rl_index_create_f16) or an fp32-stored one whose every element is an fp16 value (measured when its HI
 * image is built: max |e_lo| == 0); batches of >= 3 queries, dim % 32 == 0, dim >= 256, no empty chunk: the product of
 * two fp16 values is exact in fp32, the index stores e itself and an fp16 query has no lo half, so the ONE-product pass of SIXTEEN queries
 * (maxsim_pp.hip) already accumulates q . e in fp32 -- its exact top-k IS the result: no error bound, no candidate list, no re-scoring
 * kernel (RL_OPT_F16_EXACT = 0: the bound-filtered pipeline of rl_maxsim_topk_batch, for A/B and parity tests).  Scores: fp32-accumulated
 * dot products in the pass's summation order (integer-valued data: bit-identical to every other route; float data: within 2^-12 relative
 * of float64, like them).  A query whose elements do not survive the pass's power-of-two scaling (a dynamic range of more than 2^24 inside
 * one query) or an index with fewer than k scorable chunks raises a device flag and the full-precision passes answer the batch (as the
 * bound-filtered pipeline's fallback; rl_index_filter_stats reports kind RL_FILTER_MAXSIM_F16_EXACT, 0 candidates, the flag).
 *   query_vecs_f16 [n_queries x nq x dim] IEEE fp16 bits, 8-byte aligned; outputs as rl_maxsim_topk_batch. */
int rl_maxsim_topk_batch_f16(rl_index* index, const uint16_t* query_vecs_f16, int32_t n_queries, int32_t nq, int32_t k,
                             float* out_scores, int32_t* out_chunks, int mem, void* stream);

/* ---- rl_maxsim_topk_batch over a corpus SHARDED across several indexes, with ONE candidate threshold for all shards -----------------
 * Every shard calling rl_maxsim_topk_batch on its own re-scores the chunks near ITS k-th best approximate score: ~235 candidates per
 * query on each of eight shards of the benchmark corpus where one index re-scores 307 in all, and that work does not shrink with the
 * shard (profiles/r03_ah_shard_step_times.txt).  The threshold only needs the k-th best approximate score over ALL shards -- which lies
 * in the union of the shards' k best -- and the largest of the shards' error bounds:
 *     rl_maxsim_batch_begin(index, queries, B, nq, k, approx)        approximate passes over this shard; approx [B x (k + 1)] f32:
 *                                                                     its k best approximate scores per query (descending), then its bound m
 *     <all-gather approx over the shards: all_approx [world x B x (k + 1)]   (rl_allgather_u32 on the bits, or any all-gather)>
 *     rl_maxsim_batch_finish(index, queries, all_approx, world, rank, out_scores, out_chunks)
 *                                                                     candidates above (global k-th best) - max m - own m, re-scored exactly
 * out_* [B x k]: this shard's chunks of that candidate set, ranked by exact score (fewer than k: padded with -inf / -1); merging the shards'
 * lists (rl_allgather_merge_topk) gives bit for bit what rl_maxsim_topk_batch returns for ONE index over the whole corpus.  A shard whose
 * bound does not decide (list overflow, fewer than k scorable chunks anywhere) answers with its exact local top-k instead, on the device,
 * as the unsharded call does.  `queries` must be the same buffer contents in both calls; no other call on this index in between.
 * RL_ERR_UNSUPPORTED from _begin (no image of the hi halves on this index, fewer than three queries, a batch size with n % 8 in {1, 2},
 * nq > 32, k > 512): call rl_maxsim_topk_batch instead -- the merge accepts either. */
int rl_maxsim_batch_begin(rl_index* index, const float* query_vecs, int32_t n_queries, int32_t nq, int32_t k, float* out_approx, int mem,
                          void* stream);
int rl_maxsim_batch_finish(rl_index* index, const float* query_vecs, const float* all_approx, int32_t world, int32_t rank, float* out_scores,
                           int32_t* out_chunks, int mem, void* stream);

/* rl_maxsim_approx_scores: the FIRST stage of rl_maxsim_topk_batch's bound-filtered pipeline on its own, for tests and for
 * callers that want the bound: the approximate MaxSim score of every (query, chunk) from the hi halves of corpus and queries
 * (one fp16 MFMA product per multiply; `kernel` = 0: the sixteen-queries-per-pass kernel of maxsim_pp.hip, 1: the
 * eight-queries-per-pass kernel of maxsim_gemm.hip -- the same products and the same sums over K; the 32 per-vector maxima of a
 * chunk are added in another order, so float scores agree to the last bits, integer-valued ones exactly), and per query the rigorous bound
 * m with |approximate - exact| <= m for EVERY chunk that rl_maxsim_topk_batch's candidate window (2 m) is built on.
 *   query_vecs [n_queries x nq x dim] f32, nq <= 32;  out_scores [n_queries x n_chunks] f32 (tombstoned chunks included: no mask),
 *   out_bound [n_queries] f32 or NULL.  RL_ERR_UNSUPPORTED when the index keeps no image for the approximate pass (small / exact-fp32
 *   indexes, indexes with empty chunks). */
int rl_maxsim_approx_scores(rl_index* index, const float* query_vecs, int32_t n_queries, int32_t nq, int kernel,
                            float* out_scores, float* out_bound, int mem, void* stream);

/* rl_maxsim_rerank: the rerank shape (SURVEY.md cfg 3).  `n_queries` independent queries in one
 * launch, each with its own nq query vectors and its own list of n_cand candidate chunk ordinals.
 *   query_vecs [n_queries x nq x dim] f32, candidates [n_queries x n_cand] int32
 *   out_scores [n_queries x n_cand] f32 in candidate order (the caller sorts: the reference
 *   reorders by `result.doc_id`, src/raglite/_search.py:396).
 * A candidate of -1 (the padding of rl_search_chunks results), a tombstoned chunk, and -- for RL_MEM_DEVICE
 * callers, whose lists are not validated on the host -- any ordinal outside [0, n_chunks) scores -inf. */
int rl_maxsim_rerank(rl_index* index, const float* query_vecs, int32_t n_queries, int32_t nq,
                     const int32_t* candidates, int32_t n_cand, float* out_scores, int mem, void* stream);

/* ---- section 8e: shard merge --------------------------------------------------------------------
 * Merge `n_lists` per-shard top-k lists per query (as produced by an all-gather of each rank's
 * rl_search_rows / rl_maxsim_topk output with GLOBAL ids) into the global top-k by
 * (score desc, id asc).  in_* are [n_lists x n_queries x k_in]; out_* are [n_queries x k]. */
int rl_merge_topk(const float* in_scores, const int32_t* in_ids, int32_t n_lists, int32_t n_queries,
                  int32_t k_in, int32_t k, float* out_scores, int32_t* out_ids, int mem, void* stream);

/* The exchange step behind the C ABI: a communicator over RCCL (librccl, loaded on first use) for one process per
 * GPU.  Rank 0 calls rl_comm_unique_id and hands the 128 bytes to every rank by whatever side channel the host
 * program has (a file, an environment variable, MPI, torch.distributed's store); then EVERY rank calls rl_comm_init
 * (collective).  rl_allgather_topk: each rank's local top-k lists (device pointers, [n_queries x k]; ids LOCAL, made
 * global by adding id_offset, -1 stays -1) -> ONE ncclAllGather of (score bits, global id) records on `stream` ->
 * out_* [world x n_queries x k] on every rank.  rl_allgather_merge_topk: the same followed by rl_merge_topk's kernel:
 * out_* [n_queries x k] = the global top-k by (score desc, id asc), identical on every rank and identical to what one
 * GPU holding the whole corpus returns.  Nothing synchronises with the host.  A communicator is used by one
 * thread at a time (calls serialise on an internal mutex).  RL_ERR_UNSUPPORTED when librccl cannot be loaded.
 * A rank whose LOCAL step failed still has to enter the collective (the others are waiting in it): it contributes lists whose
 * ids are RL_ID_SHARD_MISSING (scores ignored).  rl_allgather_topk hands the marker through; rl_allgather_merge_topk answers
 * with every score NaN and every id -1 on EVERY rank -- a merge that lacks a shard never looks like an answer (no host read-back). */
#define RL_COMM_ID_BYTES 128
#define RL_ID_SHARD_MISSING (-2)
typedef struct rl_comm rl_comm;
int rl_comm_unique_id(void* out_id /* RL_COMM_ID_BYTES */);
int rl_comm_init(rl_comm** out, int rank, int world, const void* unique_id);
int rl_comm_info(const rl_comm* comm, int* rank, int* world);
int rl_comm_destroy(rl_comm* comm);
int rl_allgather_topk(rl_comm* comm, const float* local_scores, const int32_t* local_ids, int32_t n_queries, int32_t k,
                      int32_t id_offset, float* out_scores, int32_t* out_ids, void* stream);
int rl_allgather_merge_topk(rl_comm* comm, const float* local_scores, const int32_t* local_ids, int32_t n_queries,
                            int32_t k_in, int32_t id_offset, int32_t k, float* out_scores, int32_t* out_ids, void* stream);
/* The two small collectives the sharded rank cut (rl_rank_cut_*) needs, on DEVICE buffers, asynchronous on `stream`:
 * buf[i] <- sum over the ranks of buf[i] (in place), and out [world x count] <- every rank's local [count]. */
int rl_allreduce_sum_u32(rl_comm* comm, uint32_t* buf, int64_t count, void* stream);
int rl_allgather_u32(rl_comm* comm, const uint32_t* local, int64_t count, uint32_t* out, void* stream);

/* Generic exact top-k over a dense score matrix [n_queries x n] (row stride ld), the selection
 * stage used by every search above; exposed for tests and for callers that score elsewhere. */
int rl_topk(const float* scores, int32_t n_queries, int64_t n, int64_t ld, int32_t k,
            float* out_scores, int32_t* out_ids, int mem, void* stream);

/* ---- metadata filter pushed down ---------------------------------------
 * The filter-first branch of the reference's vector search (src/raglite/_search.py:96-119): only
 * rows of chunks that match the metadata filter are ranked.  The caller evaluates the filter on its
 * metadata (host) and passes the result as a bitset over chunk ordinals:
 *   chunk_filter  uint32[(n_chunks + 31) / 32], bit (c % 32) of word (c / 32) set <=> chunk c may
 *                 match (host or device per `mem`); NULL = no filter.
 * Semantics otherwise identical to the unfiltered calls; slots that cannot be filled are reported
 * as (-inf, -1) and excluded from out_counts.  Tombstoned chunks never match, filter or not. */
int rl_search_rows_filtered(rl_index* index, const float* queries, int32_t n_queries, int32_t k,
                            const uint32_t* chunk_filter, float* out_scores, int32_t* out_rows, int mem,
                            void* stream);
int rl_search_chunks_filtered(rl_index* index, const float* queries, int32_t n_queries, int32_t num_hits,
                              int32_t k, const uint32_t* chunk_filter, float* out_scores, int32_t* out_chunks,
                              int32_t* out_counts, int mem, void* stream);
int rl_maxsim_topk_filtered(rl_index* index, const float* query_vecs, int32_t nq, int32_t k,
                            const uint32_t* chunk_filter, float* out_scores, int32_t* out_chunks, int mem,
                            void* stream);

/* The order-first-then-filter branch of the reference (src/raglite/_search.py:120-141, taken when the
 * metadata filter matches more than 100 000 embedding rows): `ORDER BY dist LIMIT 1_000_000` over the
 * UNFILTERED table, then the filter, then `ORDER BY dist LIMIT num_hits`.
 *   rank_limit  > 0: per query only its rank_limit nearest live rows -- order (similarity descending,
 *               row ascending), ties on the boundary resolved to the lowest rows (SQL leaves them
 *               unspecified) -- are eligible; chunk_filter (may be NULL) is applied to those.
 *               rank_limit >= live rows, or 0, gives exactly the *_filtered result.
 * The cut is exact (three-level radix select over the score keys), not an ANN artefact. */
int rl_search_rows_ranked(rl_index* index, const float* queries, int32_t n_queries, int32_t k,
                          const uint32_t* chunk_filter, int64_t rank_limit, float* out_scores, int32_t* out_rows,
                          int mem, void* stream);
int rl_search_chunks_ranked(rl_index* index, const float* queries, int32_t n_queries, int32_t num_hits,
                            int32_t k, const uint32_t* chunk_filter, int64_t rank_limit, float* out_scores,
                            int32_t* out_chunks, int32_t* out_counts, int mem, void* stream);

/* ---- device half of update_query_adapter ----------------------------------
 * src/raglite/_query_adapter.py:153-205 fits the query adapter from evals: per eval a vector search
 * (rl_search_chunks, batched over all evals), then for every retrieved chunk the row
 * np.argmax(chunk.embedding_matrix @ q) (:174,180) as positive / negative example, then NNLS +
 * Procrustes on the host.
 *   rl_chunk_best_rows: queries [B x dim] f32, candidates [B x n_cand] int32 chunk ordinals (-1 =
 *     none) -> out_rows [B x n_cand] int32 row ordinals (first maximum on ties; -1 for no chunk or an
 *     empty chunk).
 *   rl_gather_rows: out[i] = embedding row rows[i] as f32 (NaN row for an out-of-range ordinal). */
int rl_chunk_best_rows(rl_index* index, const float* queries, int32_t n_queries, const int32_t* candidates,
                       int32_t n_cand, int32_t* out_rows, int mem, void* stream);
int rl_gather_rows(rl_index* index, const int32_t* rows, int64_t n, float* out, int mem, void* stream);

/* ---- semantic-chunking similarities ---------------------------------------
 * src/raglite/_split_chunks.py:54-72, the consumer of the pooled chunklet embeddings and the cost
 * vector of the chunk-partition MILP (which stays on the host), batched over documents:
 *   X            [n x dim] f32 chunklet embeddings of all documents, concatenated (nonzero rows)
 *   doc_offsets  int64[n_docs + 1] CSR over those rows (same side as X); NULL = one document
 *   nonoutlying  uint8[n], nonzero = chunklet size within the document's 15 %..85 % quantiles
 *                (:57-58, computed on the host from string lengths); NULL = no discourse removal
 *   out          f32[n]: out[i] = max((x_i . x_{i+1} + 1) / 2, sqrt(eps)) on the normalised,
 *                discourse-free rows (:59-72); 0 for the last row of a document. */
int rl_partition_similarity(const float* X, int64_t n, int32_t dim, const int64_t* doc_offsets, int64_t n_docs,
                            const uint8_t* nonoutlying, float* out, int mem, void* stream);

/* What the last bound-filtered search on this index did (diagnostic; bench.py reports it next to every timed number that depends
 * on it).  The searches that rank on approximate scores and re-score what a rigorous bound cannot rule out -- rl_maxsim_topk_batch
 * over the HI image, rl_search_rows for B <= 16 over the HI plane, the fused top-k of B >= 96 -- keep per-query candidate lists of a
 * fixed capacity and a device flag on which their full-precision fallback runs.  Synchronises `stream`, then fills
 *   out[0] = kind (rl_filter_kind; 0 = no such search ran since the index was created)   out[1] = queries of that call
 *   out[2] = sum of the candidate counts   out[3] = largest count   out[4] = list capacity   out[5] = 1 when the fallback ran.
 * How the data decides both numbers is the reason this exists: they are not constants of the algorithm. */
enum rl_filter_kind { RL_FILTER_NONE = 0, RL_FILTER_MAXSIM_BATCH = 1, RL_FILTER_ROWS_HI = 2, RL_FILTER_ROWS_FUSED = 3, RL_FILTER_ROWS_FUSED_HI = 4,
                      RL_FILTER_MAXSIM_F16_EXACT = 5 };
int rl_index_filter_stats(rl_index* index, int64_t out[6], void* stream);

/* Timing hook for bench.py: run `fn`-independent -- records the elapsed milliseconds between two
 * events on `stream` bracketing `iters` back-to-back launches of the named kernel path with the
 * given index / query.  kind: 0 = rl_maxsim_scores kernel only, 1 = rl_search_rows scan kernel
 * only (no selection), 2 = the two-queries-per-pass MaxSim kernel of rl_maxsim_topk_batch (query_vecs_dev
 * then holds two queries of nq / 2 vectors each; RL_ERR_UNSUPPORTED where that kernel does not apply), 3 = the
 * eight-queries-per-pass kernel over the pre-split corpus image (eight queries of nq / 8 vectors each), 4 = the
 * ranking pass of the half-bytes search of rl_search_rows (nq <= 16 queries over the fp16 HI plane;
 * RL_ERR_UNSUPPORTED when the index has none), 5 = the approximate eight-query MaxSim pass over the HI image
 * (as kind 3) with two MFMA products per multiply, 6 = the same pass with one,
 * 7 = the SIXTEEN-queries-per-pass kernel over the HI image, one product (maxsim_pp.hip: what rl_maxsim_topk_batch runs by default;
 * sixteen queries of nq / 16 vectors each; nq > 512: nq / 32 queries of 32 vectors, all their passes in ONE launch -- grid row = pass --
 * as the batch pipeline launches them), 8 = the candidate pass of the LAST rl_search_rows call of >= 96 queries on this index that
 * went through the fused top-k over the HI image, replayed with that call's queries and thresholds (query_vecs_dev / nq are not read;
 * RL_ERR_UNSUPPORTED when no such call ran or any other search used the index since), 9 = the matrix pipe alone: the sixteen-query
 * kernel's MFMA stream (8 waves per CU, 128 accumulators each) on register-resident pseudo-random fp16 operands -- no loads, no LDS, no
 * epilogue; one launch = compute units x 8 waves x 32 000 v_mfma_f32_16x16x32_f16 (x 16 384 flop): the SUSTAINED fp16 rate of this
 * device at the clock it settles at, which bench.py prints next to the nominal peak (query_vecs_dev / nq are not read),
 * 10 = the approximate pass of the few-queries MaxSim route (rl_maxsim_topk, batches of < 3; RL_OPT_HI_FEW): ONE query of nq <= 32 vectors,
 * MaxSim over the row-major fp16 HI plane (RL_ERR_UNSUPPORTED when the index has none).
 * Used so that roofline.achieved is measured with HIP
 * events on the stream the kernel runs on. */
int rl_time_kernel(rl_index* index, int kind, const float* query_vecs_dev, int32_t nq, int32_t iters,
                   float* out_ms_total, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RAGLITE_HIP_H */
