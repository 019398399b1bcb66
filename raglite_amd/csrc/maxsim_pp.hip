// a9 at batch scale, the approximate pass of the headline pipeline (api.hip: rl_maxsim_topk_batch): MaxSim of SIXTEEN queries
// (up to 32 vectors each) per corpus pass over the HI image -- the fp16 hi halves of the corpus in the one-plane image layout of
// maxsim_gemm.hip (preformat / presplit_hi_rows_kernel) -- at ONE fp16 MFMA product per multiply (q_hi . e_hi):
//
//   out[q * out_stride + c] ~ sum_{i < nq} max_{j in chunk c} Q_q[i] . D[j]      q = 0 .. n_q - 1
//
// Multi-query generalisation of src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174 behind the reranker plugin
// call src/raglite/_search.py:394-396; what it drops (the corpus' and the queries' lo halves) is bounded rigorously by the caller,
// which re-scores every chunk the bound cannot rule out with exact fp32 products (maxsim_pairs_kernel).
//
// The arrangement (DESIGN.md 4.1 has the measurements behind every point; what was tried and removed: docs/history_r04_pp_kernel.md):
//   * tile = 128 corpus rows x 512 query vectors (16 queries), K slabs of 32; 8 waves, wave w owns queries 2w and 2w + 1: 128
//     accumulator registers (2 queries x 2 blocks of 16 vectors x 8 blocks of 16 rows).  The CORPUS block is the A operand of
//     v_mfma_f32_16x16x32_f16: lane (g, n) of accumulator register u holds row 4 g + u of the block, query vector n of the register's set;
//   * the CORPUS slabs (8 KiB, `nt`) go through an LDS ring of 8 by `global_load_lds_dwordx4`, one 1-KiB piece per wave and slab, the
//     pieces of slab g + 6 issued at the top of slab g (slot (g + 6) % 8 is neither slab g's, which a lagging wave may still read, nor
//     g + 1's);
//   * the QUERY fragments (32 KiB per slab, L2-resident) go STRAIGHT TO REGISTERS: a wave owns its two queries, nobody else reads their
//     fragments.  Every wave issues four `global_load_dwordx4` at the top of slab g for slab g + 1, into the register set slab g - 1
//     multiplied from, and waits for them BY COUNT at the end of the slab (VMEM returns in order: the corpus piece is issued right AFTER the
//     query loads and stays outstanding across that wait -- an HBM piece gets two slabs to land);
//   * NO store enters a wave's VMEM queue inside the loop (it would sit in front of every later query load and takes ~2 us to retire under
//     this load): the RESULTS wait in the 96 KiB of LDS next to the ring (slot = the chunk's ordinal among the chunks the workgroup owns)
//     and are written out with coalesced stores when the workgroup is done, or the buffer nearly full;
//   * a wave's corpus fragments of slab g + 1 are read from LDS WHILE it multiplies slab g from registers -- each fragment register is
//     re-loaded right after its four MFMAs -- through inline asm, waited for by COUNT (LDS reads return in order);
//   * ONE workgroup barrier per slab, HALF-WAY through it; waves 4-7 issue every fragment read one MFMA group later than their SIMD
//     partners (LAG), so one wave's LDS instructions sit beside the partner's MFMAs instead of beside the partner's own;
//   * tile epilogue in REGISTERS (the comment in front of it has the steps): ~93 VALU instructions per 16-row block, no LDS traffic but
//     the staged results.
// Deterministic (fixed MFMA order per (query vector, row), fixed sum tree over the query vectors; independent of the grid);
// integer-valued data is exact.  Same products and the same sums over K as maxsim_gemm_kernel's one-product pass; the 32 per-vector maxima
// are added in another order: scores agree to the last bits, not bit for bit, on float data.  Needs an index without empty chunks (a chunk
// is found by counting chunk ends), nq <= 32, dim % 32 == 0, dim >= 256.
//
// MODE 1 (round 5): the SAMPLE pass of the same search on the same tile -- the similarities of every stride-th 256-row tile with all queries,
// written out (the tile epilogue applies the metric's transform; its stores sit in the wave's in-order VMEM queue in front of the next tile's
// first query loads, a stall of a few microseconds per 25-us tile that a pass of two or three tiles per workgroup can afford).  Was the
// eight-group kernel of maxsim_gemm.hip (one tile per workgroup, start-up bound): profiles/r05_*.
//
// MODE 2: the same main loop as the candidate pass of the fused exact row top-k (api.hip: search_rows_fused_hi; BASELINE cfg 5,
// src/raglite/_search.py:69-79 at B = 1000): a "query" of the tile is a GROUP of 32 single-vector queries (the fragment layout of
// query_rows_planes_kernel is that of query_planes_kernel), blockIdx.y = the query tile, and the tile epilogue compares every accumulator
// with its query's threshold -- one test per (block, column set) on the largest of a lane's four rows; for cosines against
// thr_q * min / max |e| of the register's 16-row block (a superset of the exact test) -- and appends what passes to a WAVE-PRIVATE log:
// staged in LDS, moved to global memory in coalesced bursts.  When the workgroup is done each wave re-evaluates its records with the exact
// formulas of maxsim_gemm.hip MODE 2 (same statements, same bits), keeps those that reach the threshold exactly, and appends them to the
// per-query candidate lists.  A launch covers a RANGE of row tiles: api.hip runs the pass in two rounds and tightens the thresholds between.
#include <cstdio>
#include <cstdlib>
#include <utility>

#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int PP_RT = 128;                       // corpus rows per tile
constexpr int PP_NBLK = PP_RT / 16;              // 16-row blocks per tile
constexpr int PP_QPP = 16;                       // queries per pass
constexpr int PP_DC = 8;                         // corpus ring depth (slabs)
constexpr int PP_LC = 6;                         // corpus look-ahead: the pieces of slab g + PP_LC are issued during slab g
constexpr int PP_CSLOT = PP_NBLK * 1024;         // corpus slab: 8 blocks x 1 KiB
constexpr int PP_STAGE = 1536;                   // staged results per query (MODE 0) / records per wave (MODE 2)
constexpr int PP_LDS = PP_DC * PP_CSLOT + 16 * PP_STAGE * 4;    // 64 + 96 = 160 KiB: all of a CU's LDS

__device__ __forceinline__ int64_t pp_uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// One 1-KiB piece HBM -> LDS: lane i copies 16 B from src + 16 i to lds + 16 i.  Invisible to the compiler's vmcnt bookkeeping
// (asynchronous; certified by the explicit waits below).  `nt`: streamed once.
__device__ __forceinline__ void pp_dma(uint32_t lds, const char* src, uint32_t lane16) {
    const uint32_t l = __builtin_amdgcn_readfirstlane(lds);
    const char* const p = reinterpret_cast<const char*>(pp_uniform_i64(reinterpret_cast<int64_t>(src)));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(l), "v"(lane16), "s"(p) : "memory", "m0");
}

// One LDS fragment read (64 lanes x 16 B) into 4 VGPRs, invisible to the compiler's lgkmcnt bookkeeping -- its own waits would sit in
// front of every MFMA group (the loop body is not one basic block, and the pass gives up at the joins): the value is valid after the
// counted `s_waitcnt lgkmcnt` + pp_pin().
template <int OFF>
__device__ __forceinline__ void pp_read(f32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pp_pin(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
// MODE 2: a load the compiler must not see (a compiler-visible VMEM load anywhere in the K loop makes its wait-count pass put
// `s_waitcnt vmcnt(0)` at the loop header, which drains the look-ahead DMAs on every iteration) goes through the scalar cache:
typedef float f32x16s __attribute__((ext_vector_type(16)));
// 16 floats at a 64-byte aligned address: a tile's eight (min, max) pairs / a block's sixteen row norms
__device__ __forceinline__ f32x16s pp_sload16(const float* p) {
    f32x16s v;
    const float* const u = reinterpret_cast<const float*>(pp_uniform_i64(reinterpret_cast<int64_t>(p)));
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(u) : "memory");
    return v;
}
__device__ __forceinline__ h16x8 pp_h(const f32x4& v) {
    h16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
}  // namespace

// MODE 2 arguments (see the header comment).  Queries come in groups of 32 (n_q = groups); a query tile is 16 groups.
struct PpRows {
    const float* tau;                     // [B] threshold on the similarity of query q (row_threshold_kernel: k-th of the sample - window)
    const float* q_unscale;               // [B] 2^(ex - 14) per query
    const float* q_sumsq;                 // [B] |q|^2 (cosine)
    const float* row_norm;                // [n_rows] |e| (cosine)
    const float* blk_minmax;              // [ceil(n_rows / 16)][2] min / max |e| over the rows of a 16-row block (cosine)
    int32_t B, QT, metric;                // queries, query tiles (of 512) per row tile, SCAN_COSINE | SCAN_DOT
    float* cand_scores; int32_t* cand_ids; uint32_t* cand_cnt; uint32_t* overflow; int32_t cap;  // per-query lists (as maxsim_gemm.hip MODE 2)
    uint2* log; int32_t log_cap;          // [workgroups * 8][log_cap] wave-private records (raw accumulator, packed coordinates)
    int32_t tile_begin, tile_count;       // the 128-row tiles this launch covers: [tile_begin, tile_begin + tile_count)
    // MODE 1 (the SAMPLE pass of the fused top-k): scores of every `stride`-th 256-row tile, S[q * ld_s + 256 j + r] = similarity of query q with
    // row 256 stride j + r (rows past the corpus: -inf); tile_count = 2 x the sampled 256-row tiles (a launch tile has 128 rows)
    float* S; int64_t ld_s; int32_t stride;
};

// DBG: timing skeletons (experiment builds only, -DRAGLITE_EXPERIMENTS + RAGLITE_PP_DBG / RAGLITE_PP_ROWS_DBG; WRONG results):
// 2 = no MFMAs, 8 = no LDS fragment reads, 16 = no corpus DMAs, 32 = no query loads, 128 = no block epilogues (the MFMAs stay).
// ROWNORM (MODE 2, cosine): the block-wide bound is only the PREFILTER; a group that passes it is tested row by row against T |e_row|.
template <int DBG, int MODE, bool ROWNORM>
__global__ __launch_bounds__(512, 2) void maxsim_pp_kernel(const char* __restrict__ planes, int64_t n_rows, int32_t nslab,
                                                           const char* __restrict__ qfrag, const float* __restrict__ qmeta, int32_t n_q,
                                                           const int32_t* __restrict__ row_to_chunk, const int64_t* __restrict__ chunk_offsets,
                                                           const uint32_t* __restrict__ ends_bits, float* __restrict__ out, int64_t out_stride,
                                                           float inv_e_scale, const uint32_t* __restrict__ run_if, PpRows rs) {
    constexpr bool ROWS = MODE == 2, SCORES = MODE == 1, GROUPS = ROWS || SCORES;  // GROUPS: a "query" of the tile is a group of 32 single-vector queries
    __shared__ __attribute__((aligned(16))) char smem[PP_LDS];
    if (run_if && __builtin_amdgcn_readfirstlane((int)*run_if) == 0) return;  // whole grid: a guarded launch that is not needed
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();
    const int64_t G = gridDim.x, b = blockIdx.x;
    // MODE 0: chunk-aligned row range of this workgroup (as in maxsim_gemm.hip): first chunk boundary at or after n_rows * b / G.
    auto boundary = [&](int64_t t) -> int64_t {
        if (t <= 0) return 0;
        if (t >= n_rows) return n_rows;
        const int32_t c = row_to_chunk[t];
        const int64_t c0 = chunk_offsets[c], c1 = chunk_offsets[c + 1];
        return c0 == t ? t : c1;
    };
    int32_t r_lo = 0, r_hi = 0, nt_rows = 0;
    if constexpr (!GROUPS) {
        r_lo = (int32_t)pp_uniform_i64(boundary((n_rows * b) / G));
        r_hi = (int32_t)pp_uniform_i64((b + 1 == G) ? n_rows : boundary((n_rows * (b + 1)) / G));
        if (r_hi <= r_lo) return;  // whole workgroup
    } else {
        // MODE 2: blockIdx.y = the query tile (16 groups of 32 queries), the workgroups of a grid row share the 128-row tiles evenly: a
        // workgroup multiplies ONE set of queries for its whole life, like a MaxSim pass
        const int64_t Tr = rs.tile_count;  // (of this launch: the candidate pass runs in rounds over ranges of row tiles, api.hip)
        const int64_t t0 = (Tr * b) / G, t1 = (Tr * (b + 1)) / G;
        nt_rows = (int32_t)(t1 - t0);
        if (nt_rows <= 0) return;  // whole workgroup
        r_lo = (int32_t)((rs.tile_begin + t0) * PP_RT);  // (MODE 1: the ORDINAL of the workgroup's first sample tile x 128 -- see tile_blk)
        r_hi = (int32_t)n_rows;
    }
    const int32_t org = r_lo & ~15;  // tiles start on a 16-row block of the image
    const int nt = GROUPS ? nt_rows : (r_hi - org + PP_RT - 1) / PP_RT;
    // first 16-row block of the workgroup's tile t.  MODE 1 walks SAMPLE tiles: ordinal o = org / 128 + t is the (o & 1)-th half of the
    // (o >> 1)-th sampled 256-row tile, which starts at row 256 stride (o >> 1)
    auto tile_blk = [&](int t) __attribute__((always_inline)) -> int32_t {
        if constexpr (SCORES) {
            const int32_t o = (org >> 7) + t;
            return ((o >> 1) * 2 * rs.stride + (o & 1)) * PP_NBLK;
        } else {
            return (org >> 4) + t * PP_NBLK;
        }
    };
    // query tile: MODE 2 -- 16 groups of 32 queries; MODE 0 -- the launch's passes (16 queries each) as grid rows: ONE launch for a batch's
    // passes instead of one per pass, so a pass's workgroups start on the CUs the previous pass's leave instead of behind a launch boundary
    const int qt = (int)blockIdx.y;
    const int q_base = GROUPS ? 0 : PP_QPP * qt;  // first query of this workgroup's pass
    const int total = nt * nslab;               // K slabs this workgroup streams
    const int32_t last_blk = (int32_t)((n_rows + 15) >> 4) - 1;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;

    // ---- this wave's queries: 2 wv and 2 wv + 1 ---------------------------------------------------------------------------------------
    const bool has0 = q_base + 2 * wv < n_q, has1 = q_base + 2 * wv + 1 < n_q;  // wave-uniform
    // hi fragments of block 0 at K slab 0 (block 1: + 2 KiB; K slab s: + 4 KiB s)
    int ql0 = PP_QPP * qt + 2 * wv, ql1 = PP_QPP * qt + 2 * wv + 1;
    ql0 = ql0 < n_q ? ql0 : n_q - 1;  // (queries the pass does not have: a valid one's fragments, never emitted)
    ql1 = ql1 < n_q ? ql1 : n_q - 1;
    const char* const qp0 = qfrag + pp_uniform_i64((int64_t)ql0 * nslab * 4096);
    const char* const qp1 = qfrag + pp_uniform_i64((int64_t)ql1 * nslab * 4096);
    int qr_s = 0;  // K slab of the NEXT query-fragment load
    auto issue_qregs = [&](f32x4 (&qn)[4]) __attribute__((always_inline)) {
        if constexpr (!(DBG & 32)) {
            const char* const a0 = reinterpret_cast<const char*>(pp_uniform_i64(reinterpret_cast<int64_t>(qp0 + (int64_t)qr_s * 4096)));
            const char* const a1 = reinterpret_cast<const char*>(pp_uniform_i64(reinterpret_cast<int64_t>(qp1 + (int64_t)qr_s * 4096)));
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:2048"
                         : "=&v"(qn[0]), "=&v"(qn[1]) : "v"(lane16), "s"(a0) : "memory");
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:2048"
                         : "=&v"(qn[2]), "=&v"(qn[3]) : "v"(lane16), "s"(a1) : "memory");
            if (++qr_s == nslab) qr_s = 0;
        }
    };
    // ---- feeder duty: every wave fetches corpus block wv of slab g + PP_LC during slab g ------------------------------------------------
    int fc_s = 0, fc_r = 0, fc_slot = 0;  // K slab and tile of the NEXT fetch, its ring slot
    const int32_t blk_org = (org >> 4) + wv;
    const int64_t slab_bytes = (int64_t)nslab * 1024;
    auto issue_c = [&]() __attribute__((always_inline)) {
        if constexpr (!(DBG & 16)) {
            const int t = fc_r < nt ? fc_r : nt - 1;  // (after the last tile: some valid tile, multiplied into sums nobody reads)
            int32_t blk = SCORES ? tile_blk(t) + wv : blk_org + t * PP_NBLK;
            blk = blk < last_blk ? blk : last_blk;  // past the image: harmless re-read of the last block, never emitted
            const char* src = planes + pp_uniform_i64((int64_t)blk * slab_bytes + (int64_t)fc_s * 1024);
            pp_dma(lds_base + (uint32_t)(fc_slot * PP_CSLOT + wv * 1024), src, lane16);
        }
        if (++fc_s == nslab) { fc_s = 0; ++fc_r; }
        fc_slot = fc_slot + 1 == PP_DC ? 0 : fc_slot + 1;
    };

    // ---- accumulators: S[corpus row 16 a + 4 g + u][query vector 16 qb + n], lane = 16 g + n; [query of the wave][qb][a] ------
    f32x4 acc[2][2][PP_NBLK];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int a = 0; a < PP_NBLK; ++a) acc[q][qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment registers ---------------------------------------------------------------------------------------------------
    f32x4 ef[PP_NBLK];      // corpus blocks of the slab being multiplied (each re-loaded for the next slab right after its MFMAs)
    f32x4 qA[4], qB[4];     // query fragments [2 * query of the wave + qb], two sets alternating over slabs
    int c_slot = 0;         // ring slot of the slab whose fragments are READ next
    const uint32_t rd_c = lds_base + lane16;
    auto read_slab = [&](auto A_) __attribute__((always_inline)) {  // corpus block a of the next slab
        constexpr int a = decltype(A_)::value;
        if constexpr (!(DBG & 8)) pp_read<a * 1024>(ef[a], rd_c + (uint32_t)(c_slot * PP_CSLOT));
    };
    auto mfma_group = [&](f32x4 (&q)[4], auto A_) __attribute__((always_inline)) {
        constexpr int A = decltype(A_)::value;
        if constexpr (!(DBG & 2)) {
            acc[0][0][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(ef[A]), pp_h(q[0]), acc[0][0][A], 0, 0, 0);
            acc[0][1][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(ef[A]), pp_h(q[1]), acc[0][1][A], 0, 0, 0);
            acc[1][0][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(ef[A]), pp_h(q[2]), acc[1][0][A], 0, 0, 0);
            acc[1][1][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(ef[A]), pp_h(q[3]), acc[1][1][A], 0, 0, 0);
        }
    };

    // ---- block epilogues ---------------------------------------------------------------------------------------------------------------
    // MODE 0, all in registers.  A block's 16 x 64 scores sit in 16 registers: register (c = 2 q + qb, u), lane (g, n) = row 4 g + u, vector n
    // of set c:
    //   (1) 4 x 4 transposes between the register index c and the lane group g (v_permlane32_swap / v_permlane16_swap, 16 instructions):
    //       afterwards lane (G, n) owns query vector n of set G and register 4 c + u is corpus row 4 c + u -- every row of the block in one lane;
    //   (2) the per-chunk maximum is a running v_max down the 16 registers; a chunk end is the same row in every lane, so "restart here"
    //       is EXEC = 0 for that one instruction (s_bitcmp1 + s_cselect on the scalar unit): 16 VALU operations for all 64 query vectors.
    //       The chunk still open at the end of the block is register 15, carried in `run` across blocks and tiles;
    //   (3) the sum over the 32 vectors of a query, for all 16 rows at once: the two sets of a query are neighbouring lane groups (one
    //       v_permlane16_swap + add per PAIR of rows leaves rows r, r + 8 of both queries in the four lane groups), then a reduction over the 16
    //       lanes of a DPP row in which every step also halves the number of registers (bank-masked v_add_f32_dpp at distances 8 and 4,
    //       quad permutes for 2 and 1): 32 instructions, one register of results -- lane 16 G + 4 b + t holds row
    //       (t >> 1) + 2 (b & 1) + 4 (b >> 1) + 8 (G & 1) of query G >> 1 -- staged in LDS for all the chunks that end in the block.
    const int fG = lane >> 4, fb = (lane >> 2) & 3, ft = lane & 3;
    const int my_row = (ft >> 1) + 2 * (fb & 1) + 4 * (fb >> 1) + 8 * (fG & 1);
    const uint32_t my_bit = 1u << my_row, my_below = my_bit - 1u;
    float run = -INFINITY;  // maximum over the rows of the chunk still open, for this lane's query vector; lives across blocks and tiles
    uint32_t prev_last_end = 1;  // the row before the workgroup's first block closes a chunk as far as this workgroup is concerned
    // chunk ordinal of the next chunk to finish: no chunk is empty, so it advances by one per chunk end -- one scalar load per workgroup
    int32_t ord_run = GROUPS ? 0 : __builtin_amdgcn_readfirstlane(row_to_chunk[org]);
    const int e_q = fG >> 1;
    const bool e_has = !GROUPS && (e_q == 0 ? has0 : has1) && (ft & 1) == 0;  // (lanes t and t ^ 1 hold the same sum)
    const float e_unscale = e_has ? qmeta[2 * (q_base + 2 * wv + e_q)] * inv_e_scale : 0.f;
    // the results wait in LDS -- slot = the chunk's ordinal among the chunks this workgroup OWNS (they end inside [r_lo, r_hi): consecutive
    // ordinals from ord_lo on) minus what has been flushed
    [[maybe_unused]] float* const o_buf = reinterpret_cast<float*>(smem + PP_DC * PP_CSLOT) + (2 * wv + e_q) * PP_STAGE;
    [[maybe_unused]] const int32_t ord_lo = !GROUPS ? __builtin_amdgcn_readfirstlane(row_to_chunk[r_lo]) : 0;
    [[maybe_unused]] int32_t own_cnt = 0, own_flushed = 0;  // owned chunks finished so far / already written out
    [[maybe_unused]] auto flush_out = [&]() __attribute__((always_inline)) {
        if constexpr (!GROUPS) {
            const int n = own_cnt - own_flushed;  // (wave-uniform)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own LDS writes
            const float* const b0 = reinterpret_cast<const float*>(smem + PP_DC * PP_CSLOT) + (2 * wv) * PP_STAGE;
            float* const g0 = out + (int64_t)(q_base + 2 * wv) * out_stride + ord_lo + own_flushed;
            if (has0)
                for (int i = lane; i < n; i += 64) g0[i] = b0[i];
            if (has1)
                for (int i = lane; i < n; i += 64) g0[out_stride + i] = b0[PP_STAGE + i];
            own_flushed = own_cnt;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (rare: once per workgroup on RAGLite-shaped chunks)
        }
    };
    const uint64_t odd_pairs = 0xccccccccccccccccull;  // lanes with t >= 2
    auto swap32 = [](float& x, float& y) __attribute__((always_inline)) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    };
    auto swap16 = [](float& x, float& y) __attribute__((always_inline)) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        x = __uint_as_float(r[0]);
        y = __uint_as_float(r[1]);
    };
    auto zero_block = [&](auto A_) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) acc[q][qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    [[maybe_unused]] auto block_epilogue_maxsim = [&](auto A_, int T) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
        const int32_t base = org + T * PP_RT + 16 * a;
        // "last row of its chunk" bits of the block's 16 rows: half a word of the bitmap
        const uint32_t E = (ends_bits[base >> 5] >> (base & 16)) & 0xffffu;
        float z[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = acc[r >> 3][(r >> 2) & 1][a][r & 3];
        // (1)
#pragma unroll
        for (int u = 0; u < 4; ++u) { swap32(z[u], z[8 + u]); swap32(z[4 + u], z[12 + u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { swap16(z[u], z[4 + u]); swap16(z[8 + u], z[12 + u]); }
        // (2)
        uint64_t saved;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_bitcmp1_b32 %[ple], 0\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z0], %[z0], %[run]\n\t"
            "s_bitcmp1_b32 %[E], 0\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z1], %[z1], %[z0]\n\t"
            "s_bitcmp1_b32 %[E], 1\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z2], %[z2], %[z1]\n\t"
            "s_bitcmp1_b32 %[E], 2\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z3], %[z3], %[z2]\n\t"
            "s_bitcmp1_b32 %[E], 3\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z4], %[z4], %[z3]\n\t"
            "s_bitcmp1_b32 %[E], 4\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z5], %[z5], %[z4]\n\t"
            "s_bitcmp1_b32 %[E], 5\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z6], %[z6], %[z5]\n\t"
            "s_bitcmp1_b32 %[E], 6\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z7], %[z7], %[z6]\n\t"
            "s_bitcmp1_b32 %[E], 7\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z8], %[z8], %[z7]\n\t"
            "s_bitcmp1_b32 %[E], 8\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z9], %[z9], %[z8]\n\t"
            "s_bitcmp1_b32 %[E], 9\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z10], %[z10], %[z9]\n\t"
            "s_bitcmp1_b32 %[E], 10\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z11], %[z11], %[z10]\n\t"
            "s_bitcmp1_b32 %[E], 11\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z12], %[z12], %[z11]\n\t"
            "s_bitcmp1_b32 %[E], 12\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z13], %[z13], %[z12]\n\t"
            "s_bitcmp1_b32 %[E], 13\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z14], %[z14], %[z13]\n\t"
            "s_bitcmp1_b32 %[E], 14\n\ts_cselect_b64 exec, 0, -1\n\tv_max_f32 %[z15], %[z15], %[z14]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [sv] "=&s"(saved), [z0] "+v"(z[0]), [z1] "+v"(z[1]), [z2] "+v"(z[2]), [z3] "+v"(z[3]), [z4] "+v"(z[4]), [z5] "+v"(z[5]),
              [z6] "+v"(z[6]), [z7] "+v"(z[7]), [z8] "+v"(z[8]), [z9] "+v"(z[9]), [z10] "+v"(z[10]), [z11] "+v"(z[11]), [z12] "+v"(z[12]),
              [z13] "+v"(z[13]), [z14] "+v"(z[14]), [z15] "+v"(z[15])
            : [run] "v"(run), [E] "s"(E), [ple] "s"(prev_last_end)
            : "scc");
        run = z[15];  // (used only if row 15 does not end a chunk: prev_last_end)
        prev_last_end = (E >> 15) & 1u;
        // (3) rows r and r + 8 of both queries: lane groups (0: r of query 0, 1: r + 8 of query 0, 2: r of query 1, 3: r + 8 of query 1)
        float s8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            swap16(z[k], z[k + 8]);
            s8[k] = z[k] + z[k + 8];
        }
        float t0, t1, w;
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %[s0], %[s0], %[s0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[s1], %[s1], %[s1] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[s2], %[s2], %[s2] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[s3], %[s3], %[s3] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[s0], %[s4], %[s4] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %[s1], %[s5], %[s5] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %[s2], %[s6], %[s6] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %[s3], %[s7], %[s7] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %[s0], %[s0], %[s0] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %[s1], %[s1], %[s1] row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %[s0], %[s2], %[s2] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %[s1], %[s3], %[s3] row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[t0], %[s0], %[s0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %[t1], %[s1], %[s1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_e64 %[w], %[t0], %[t1], %[odd]\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[w], %[w], %[w] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
            : [s0] "+v"(s8[0]), [s1] "+v"(s8[1]), [s2] "+v"(s8[2]), [s3] "+v"(s8[3]), [t0] "=&v"(t0), [t1] "=&v"(t1), [w] "=&v"(w)
            : [s4] "v"(s8[4]), [s5] "v"(s8[5]), [s6] "v"(s8[6]), [s7] "v"(s8[7]), [odd] "s"(odd_pairs));
        int32_t lo = r_lo - base, hi = r_hi - base;  // rows of this block inside the workgroup's range
        lo = lo < 0 ? 0 : lo;
        hi = hi < 0 ? 0 : hi;
        asm("" : "+s"(lo), "+s"(hi));  // (stay on the scalar unit: the clamp is a v_med3 otherwise)
        lo = lo > 16 ? 16 : lo;
        hi = hi > 16 ? 16 : hi;
        uint32_t EM = E & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        asm("" : "+s"(EM));
        if (EM != 0u) {  // wave-uniform: some chunk of this workgroup ends in this block
            if ((EM & my_bit) && e_has) o_buf[ord_run - ord_lo - own_flushed + __builtin_popcount(E & my_below)] = w * e_unscale;
            own_cnt += __builtin_popcount(EM);
        }
        ord_run += __builtin_popcount(E);
        zero_block(A_);
    };

    // MODE 2: threshold test of every accumulator of the block, records into the wave-private log.  Register (c = 2 q + qb, a, u), lane
    // (g, n) = corpus row 16 a + 4 g + u of the tile against query 32 (16 qt + 2 wv + q) + 16 qb + n.  T4[c]: a lower bound of what the raw
    // accumulator must reach (cosine: divided by the row's |e|), per lane, computed once per workgroup: the statements of maxsim_gemm.hip's
    // epilogue_cand (1e-5 of slack for the roundings of 1 - (1 - c); thresholds of queries past B are +inf).  Cosine: acc >= T |e_row| follows
    // from acc >= min(T min|e|, T max|e|) over the row's 16-row block.
    // The four thresholds of a lane live in ONE register across the K loop (T4x: lane group g holds the threshold of column set c = g; the
    // block epilogue hands every lane its four values back with ds_bpermute, the LDS crossbar without any LDS memory) -- as four registers
    // they were what the allocator spilled, and a scratch reload inside the loop puts `s_waitcnt vmcnt(0)` at its header.
    [[maybe_unused]] float T4x = INFINITY;
    [[maybe_unused]] uint32_t ncand = 0;  // wave-uniform: records this wave has logged
    // The records wait in LDS (PP_STAGE per wave) and go to the wave's log in global memory in coalesced bursts -- when the buffer is half
    // full at the end of a tile, and when the workgroup is done.  A tile that logs more than the buffer's free half loses records: the
    // overflow flag, the dense path decides (as for a full global log).
    constexpr uint32_t LOG_CAP = PP_STAGE;
    [[maybe_unused]] uint2* const l_log = reinterpret_cast<uint2*>(smem + PP_DC * PP_CSLOT) + wv * LOG_CAP;
    [[maybe_unused]] uint32_t ncand_out = 0;  // records already moved to the global log
    [[maybe_unused]] uint2* const my_log = ROWS ? rs.log + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wv) * (size_t)rs.log_cap : nullptr;
    [[maybe_unused]] const bool cosine = GROUPS && rs.metric == SCAN_COSINE;
    // MODE 1: the lane's four unscale factors and four 1 / |q| live in ONE register each across the K loop, like T4x (lane group g holds
    // column set c = g's value; the tile epilogue hands them back with ds_bpermute)
    [[maybe_unused]] float Ux = 0.f, Rx = 0.f;
    if constexpr (SCORES) {  // (before the first DMA: these loads are ordinary ones)
        const int c = lane >> 4;
        const int32_t q = (16 * qt + 2 * wv + (c >> 1)) * 32 + 16 * (c & 1) + (lane & 15);
        const int32_t qc = q < rs.B ? q : rs.B - 1;
        Ux = q < rs.B ? rs.q_unscale[qc] * inv_e_scale : 0.f;  // (0: a query past B -- its column is not stored)
        Rx = cosine ? 1.0f / sqrtf(rs.q_sumsq[qc]) : 1.0f;
    }
    if constexpr (ROWS) {  // (before the first DMA: these loads are ordinary ones)
        const int c = lane >> 4;
        const int32_t q = (16 * qt + 2 * wv + (c >> 1)) * 32 + 16 * (c & 1) + (lane & 15);
        const int32_t qc = q < rs.B ? q : rs.B - 1;
        const float us = rs.q_unscale[qc] * inv_e_scale;  // > 0, a power of two
        const float tq = rs.tau[qc];
        const bool bad = q < rs.B && !(tq > -INFINITY);  // NaN or -inf: unusable
        const float slack = 1e-5f * fmaxf(1.0f, fabsf(tq));
        const float t_dot = cosine ? (tq - slack) * sqrtf(rs.q_sumsq[qc]) : (tq - 1.0f) - slack;  // bound on d (cosine: on d / |e|)
        T4x = q < rs.B ? (t_dot - fabsf(t_dot) * 1e-6f) / us : INFINITY;
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *rs.overflow = 1u;
    }
    // tq4: the lane's four thresholds (T4x handed back by rows_thresholds()); nmin / nmax: the block's norm range (cosine)
    [[maybe_unused]] auto rows_thresholds = [&](float (&tq4)[4]) __attribute__((always_inline)) {
        // (the four lane addresses are re-derived per call from one opaque value: hoisted out of the K loop they are four registers, one of
        // which the allocator spills -- and a scratch reload brings `s_waitcnt vmcnt(0)` with it)
        uint32_t b;  // the lane number, re-read here (volatile: not hoisted), -> (lane & 15) << 2
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(b));
        b = (b & 15u) << 2;
#pragma unroll
        for (int c = 0; c < 4; ++c)  // column set c's thresholds sit in lane group c of T4x
            tq4[c] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(((uint32_t)c << 6) | b), __float_as_int(T4x)));
    };
    [[maybe_unused]] auto block_epilogue_rows = [&](auto A_, int T, const float (&tq4)[4], float nmin, float nmax) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
        const int32_t base = org + T * PP_RT + 16 * a;
        if (base < (int32_t)n_rows) {  // (wave-uniform) blocks past the corpus are re-reads of its last block: not scored
            // Opaque copy of the lane's coordinates: with the plain value every record word below is loop-invariant, the compiler hoists
            // them out of the K loop and spills them (the note in maxsim_gemm.hip's epilogue_cand)
            uint32_t lane_code = ((uint32_t)(lane & 15) << 7) | (uint32_t)(4 * (lane >> 4));  // (query column, first row of the lane's quad)
            asm volatile("" : "+v"(lane_code));
            float t4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) t4[c] = tq4[c];
            if (cosine) {
                // two scalars per block, slightly widened: T * |e| is formed with one rounding here and none in the exact test
                nmin *= 0.999999f;
                nmax *= 1.000001f;
#pragma unroll
                for (int c = 0; c < 4; ++c) t4[c] = fminf(t4[c] * nmin, t4[c] * nmax);
            }
            const uint32_t code_a = ((uint32_t)T << 13) | (uint32_t)(16 * a);
            if constexpr (ROWNORM) {
                // ROW-NORM variant (api.hip launches it for cosines over an index whose row norms span more than a factor of four): the
                // block-wide bound is only the PREFILTER; a group that passes it is tested row by row against T * |e_row| -- the block's
                // sixteen norms by ONE scalar load, only in blocks where some group gets that far.  With norms that differ wildly inside a
                // block the block-wide bound passes nearly everything, the record logs fill and the dense path has to answer; on unit-norm
                // corpora it costs 4 % for nothing (profiles/r04_ag_*), hence two instantiations.  The corpus' last, partial block keeps
                // the block-wide bound (its norms end with the array).
                float top4[4];
                bool any_hit = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 v4 = acc[c >> 1][c & 1][a];
                    top4[c] = fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3]));
                    any_hit |= __builtin_amdgcn_ballot_w64(top4[c] >= t4[c]) != 0ull;
                }
                float nv0 = 1.f, nv1 = 1.f, nv2 = 1.f, nv3 = 1.f;
                const bool row_norms = cosine && base + 16 <= (int32_t)n_rows;  // (wave-uniform)
                if (any_hit && row_norms) {
                    const f32x16s nr = pp_sload16(rs.row_norm + base);
                    const int gq = lane >> 4;
                    nv0 = gq == 0 ? nr[0] : gq == 1 ? nr[4] : gq == 2 ? nr[8] : nr[12];
                    nv1 = gq == 0 ? nr[1] : gq == 1 ? nr[5] : gq == 2 ? nr[9] : nr[13];
                    nv2 = gq == 0 ? nr[2] : gq == 1 ? nr[6] : gq == 2 ? nr[10] : nr[14];
                    nv3 = gq == 0 ? nr[3] : gq == 1 ? nr[7] : gq == 2 ? nr[11] : nr[15];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 v4 = acc[c >> 1][c & 1][a];
                    if (__builtin_amdgcn_ballot_w64(top4[c] >= t4[c]) != 0ull) {  // (wave-uniform)
                        float r0 = v4[0], r1 = v4[1], r2 = v4[2], r3 = v4[3];
                        const float tc = tq4[c];
                        float n0 = nv0, n1 = nv1, n2 = nv2, n3 = nv3;
                        if (cosine && !row_norms) n0 = n1 = n2 = n3 = tc >= 0.f ? nmin : nmax;  // (nmin / nmax carry their 1e-6 already)
                        uint32_t code = (code_a + ((uint32_t)c << 11)) | lane_code;
#pragma nounroll
                        for (int u = 0; u < 4; ++u) {
                            const float tn = tc * n0;  // T |e_row|, widened by 1e-6 (one rounding here, none in the exact test)
                            const bool pass = r0 >= fminf(tn * 0.999999f, tn * 1.000001f);
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                            if (mask != 0ull) {
                                const uint32_t pos = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                                if (pass && pos - ncand_out < LOG_CAP) l_log[pos - ncand_out] = make_uint2(__float_as_uint(r0), code);
                                ncand += (uint32_t)__builtin_popcountll(mask);
                            }
                            r0 = r1; r1 = r2; r2 = r3;
                            n0 = n1; n1 = n2; n2 = n3;
                            ++code;
                            asm volatile("" : "+v"(r0), "+v"(code), "+v"(n0));  // (keeps the loop a loop)
                        }
                    }
                }
            } else {
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // one test per (block, query column) on the largest of the lane's four rows; the rare group that has a hit (~0.2 % of the
                    // scores reach a threshold: about every other group of 256) is looked at register by register -- in a LOOP of four trips
                    // that rotates the four values through one register (this code exists once per block, slab parity and wave role: unrolled
                    // it alone would be larger than the instruction cache)
                    const f32x4 v4 = acc[c >> 1][c & 1][a];
                    const float top = fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3]));
                    if (__builtin_amdgcn_ballot_w64(top >= t4[c]) != 0ull) {  // (wave-uniform)
                        float r0 = v4[0], r1 = v4[1], r2 = v4[2], r3 = v4[3];
                        uint32_t code = (code_a + ((uint32_t)c << 11)) | lane_code;
    #pragma nounroll
                        for (int u = 0; u < 4; ++u) {
                            const bool pass = r0 >= t4[c];
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
                            if (mask != 0ull) {
                                const uint32_t pos = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                                if (pass && pos - ncand_out < LOG_CAP) l_log[pos - ncand_out] = make_uint2(__float_as_uint(r0), code);
                                ncand += (uint32_t)__builtin_popcountll(mask);
                            }
                            r0 = r1; r1 = r2; r2 = r3;
                            ++code;
                            asm volatile("" : "+v"(r0), "+v"(code));  // (keeps the loop a loop)
                        }
                    }
                }
            }
        }
        zero_block(A_);
    };
    // MODE 1: similarities of the block's 16 rows with the wave's 64 queries, written out.  Register (c, a, u), lane (g, n) = row 16 a + 4 g + u
    // of the tile against query 32 (16 qt + 2 wv + (c >> 1)) + 16 (c & 1) + n: a lane's four rows of one query are 16 contiguous bytes of S.
    // cosine: 1 - (1 - d / (|e| |q|)) (scan.hip's form); dot: 1 + d; rows past the corpus: -inf.  The 16 row norms of a block come by ONE
    // scalar load (no compiler-visible VMEM load may sit in the K loop: its wait would drain the look-ahead DMAs on every slab).
    [[maybe_unused]] auto block_epilogue_scores = [&](auto A_, int T, const float (&us4)[4], const float (&rq4)[4], uint32_t n_op) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
        const int32_t o = (org >> 7) + T;                                           // sample-tile ordinal
        const int32_t base = (tile_blk(T) + a) * 16;                                // first corpus row of the block
        const int64_t col = (int64_t)(o >> 1) * 256 + (o & 1) * 128 + 16 * a;       // its column in S
        const int gq = lane >> 4;
        float rn[4] = {1.f, 1.f, 1.f, 1.f};                                         // 1 / |e| of the lane's four rows (cosine)
        if (cosine && base < (int32_t)n_rows) {                                     // (wave-uniform; the norm array is padded to whole blocks)
            const f32x16s nr = pp_sload16(rs.row_norm + base);
            rn[0] = 1.0f / (gq == 0 ? nr[0] : gq == 1 ? nr[4] : gq == 2 ? nr[8] : nr[12]);
            rn[1] = 1.0f / (gq == 0 ? nr[1] : gq == 1 ? nr[5] : gq == 2 ? nr[9] : nr[13]);
            rn[2] = 1.0f / (gq == 0 ? nr[2] : gq == 1 ? nr[6] : gq == 2 ? nr[10] : nr[14]);
            rn[3] = 1.0f / (gq == 0 ? nr[3] : gq == 1 ? nr[7] : gq == 2 ? nr[11] : nr[15]);
        }
        const int32_t left = (int32_t)n_rows - (base + 4 * gq);                     // rows of the lane's quad inside the corpus
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v4 = acc[c >> 1][c & 1][a];
            f32x4 o4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = v4[u] * us4[c];
                float v = cosine ? 1.0f - (1.0f - d * rn[u] * rq4[c]) : 1.0f + d;
                o4[u] = u < left ? v : -INFINITY;
            }
            // (n_op: the lane's column, re-read per tile -- with the plain value the four S addresses of a lane are loop-invariant, get hoisted
            // out of the K loop and spilled)
            const int32_t q = (16 * qt + 2 * wv + (c >> 1)) * 32 + 16 * (c & 1) + (int32_t)n_op;
            if (q < rs.B) *reinterpret_cast<f32x4*>(rs.S + (int64_t)q * rs.ld_s + col + 4 * gq) = o4;
        }
        zero_block(A_);
    };
    [[maybe_unused]] auto dump_log = [&]() __attribute__((always_inline)) {  // LDS -> this wave's log in global memory
        if constexpr (ROWS) {
            uint32_t n = ncand - ncand_out;  // (wave-uniform)
            if (n > LOG_CAP) {  // a tile logged more than the buffer held: records were lost
                if (lane == 0) *rs.overflow = 1u;
                n = LOG_CAP;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t i = lane; i < n; i += 64)
                if (ncand_out + i < (uint32_t)rs.log_cap) my_log[ncand_out + i] = l_log[i];
            ncand_out = ncand;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    // When the workgroup's tiles are done: every record once more with the exact formulas (maxsim_gemm.hip flush_candidates: same
    // statements, same bits), kept if it reaches its query's threshold exactly, appended to the query's list.
    [[maybe_unused]] auto flush_rows = [&]() __attribute__((always_inline)) {
        if constexpr (ROWS) {
            if (ncand > (uint32_t)rs.log_cap) {  // (wave-uniform) more than the log holds: let the dense path decide
                if (lane == 0) *rs.overflow = 1u;
                ncand = (uint32_t)rs.log_cap;
            }
            const int mode = rs.metric;
            for (uint32_t i = lane; i < ncand; i += 64) {
                const uint2 rec = my_log[i];
                const uint32_t m = rec.y;
                const int T = (int)(m >> 13), c = (int)((m >> 11) & 3u), n = (int)((m >> 7) & 15u), r = (int)(m & 127u);
                const int32_t q = (16 * qt + 2 * wv + (c >> 1)) * 32 + 16 * (c & 1) + n, row = org + T * PP_RT + r;
                if (q >= rs.B || row >= (int32_t)n_rows) continue;
                const float d = __uint_as_float(rec.x) * (rs.q_unscale[q] * inv_e_scale);  // the dot product, as MODE 1 of maxsim_gemm.hip forms it
                float v = d;
                if (mode == SCAN_COSINE) v = 1.0f - (1.0f - d / (rs.row_norm[row] * sqrtf(rs.q_sumsq[q])));
                else if (mode == SCAN_DOT) v = 1.0f + d;
                if (!(v >= rs.tau[q])) continue;  // (the block-wide cosine test is a superset)
                const uint32_t slot = atomicAdd(rs.cand_cnt + q, 1u);
                if (slot < (uint32_t)rs.cap) {
                    rs.cand_scores[(int64_t)q * rs.cap + slot] = v;
                    rs.cand_ids[(int64_t)q * rs.cap + slot] = row;
                } else {
                    *rs.overflow = 1u;
                }
            }
        }
    };

    // ---- one K slab: MFMAs of slab g from registers, and after each block's four MFMAs the LDS read that re-loads its fragment register for
    // slab g + 1.  HALF-WAY through, the workgroup barrier: with every wave's count-wait for its query fragments at the end of the previous slab
    // (which retired its corpus piece of two slabs ago) it certifies slab g + 2 as landed and slab g's LDS slot as free.  No compiler-visible
    // memory LOAD in here: the body carries no wait but the ones written out.
    // The two waves of a SIMD (w and w + 4) run the same stream half a step apart: waves 4-7 (LAG) issue each fragment read one MFMA group later
    // than waves 0-3, so that one wave's LDS instructions sit beside its partner's four MFMAs instead of beside the partner's own reads.
    // The eight block epilogues of a tile run back to back at the end of its last slab (a runtime block index -- one copy of the epilogue code
    // for all eight blocks -- does not work: the accumulators would have to be indexed at run time, which puts them in scratch memory, or be
    // merged over eight predecessors, which the register allocator answers with hundreds of spills: both tried, round 4).
    int c_s = 0, c_r = 0;  // K slab and tile of the slab being multiplied
#define PP_I(N) std::integral_constant<int, N>{}
    auto tile_end = [&]() __attribute__((always_inline)) {
        if constexpr (DBG & 128) return;
        if (c_s == nslab - 1) {  // the tile is complete: its eight blocks back to back
            if constexpr (SCORES) {
                float us4[4], rq4[4];
                uint32_t bl;  // the lane number, re-read here (volatile: not hoisted), -> (lane & 15) << 2
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(bl));
                bl = (bl & 15u) << 2;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    us4[c] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(((uint32_t)c << 6) | bl), __float_as_int(Ux)));
                    rq4[c] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)(((uint32_t)c << 6) | bl), __float_as_int(Rx)));
                }
                [&]<int... A>(std::integer_sequence<int, A...>) {
                    (block_epilogue_scores(std::integral_constant<int, A>{}, c_r, us4, rq4, bl >> 2), ...);
                }(std::make_integer_sequence<int, PP_NBLK>{});
            } else if constexpr (ROWS) {
                // ONE scalar load and one LDS round trip per tile: the eight (min, max) norm pairs (64 B) and the lane's four
                // thresholds -- a wait per block would expose a scalar-cache miss eight times per tile (the array has 625 KB)
                float tq4[4];
                rows_thresholds(tq4);
                f32x16s mm = {};
                if (cosine) mm = pp_sload16(rs.blk_minmax + 2 * ((org + c_r * PP_RT) >> 4));
                [&]<int... A>(std::integer_sequence<int, A...>) {
                    (block_epilogue_rows(std::integral_constant<int, A>{}, c_r, tq4, mm[2 * A], mm[2 * A + 1]), ...);
                }(std::make_integer_sequence<int, PP_NBLK>{});
                if (ncand - ncand_out > LOG_CAP / 2) dump_log();
            } else {
                [&]<int... A>(std::integer_sequence<int, A...>) { (block_epilogue_maxsim(std::integral_constant<int, A>{}, c_r), ...); }
                (std::make_integer_sequence<int, PP_NBLK>{});
                if (own_cnt - own_flushed > PP_STAGE - PP_RT) flush_out();  // (a tile finishes at most PP_RT chunks)
            }
        }
    };
    auto slab = [&](f32x4 (&q)[4], f32x4 (&qn)[4], auto LAG_) __attribute__((always_inline)) {
        constexpr bool LAG = decltype(LAG_)::value;
        auto G = [&](auto A_) __attribute__((always_inline)) { mfma_group(q, A_); };
        auto R = [&](auto A_) __attribute__((always_inline)) { read_slab(A_); };
        // the query fragments of the next slab, then the corpus piece of slab g + PP_LC
        issue_qregs(qn);
        issue_c();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAG) {
            G(PP_I(0)); R(PP_I(0)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(1)); R(PP_I(1)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(2)); R(PP_I(2)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(3)); R(PP_I(3)); __builtin_amdgcn_sched_barrier(0);
            // the reads of the previous slab's second half: corpus blocks 4-7 of slab g (issued half a slab ago)
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
            G(PP_I(0)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(1)); R(PP_I(0)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(2)); R(PP_I(1)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(3)); R(PP_I(2)); __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        }
        pp_pin(ef[4], ef[5], ef[6], ef[7]);
        asm volatile("s_barrier" ::: "memory");  // slab g + 2 is readable from the next slab on; everybody finished reading slab g (its last reads were waited for just above)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAG) {
            G(PP_I(4)); R(PP_I(4)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(5)); R(PP_I(5)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(6)); R(PP_I(6)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(7)); R(PP_I(7)); __builtin_amdgcn_sched_barrier(0);
        } else {
            G(PP_I(4)); R(PP_I(3)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(5)); R(PP_I(4)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(6)); R(PP_I(5)); __builtin_amdgcn_sched_barrier(0);
            G(PP_I(7)); R(PP_I(6)); __builtin_amdgcn_sched_barrier(0);
            R(PP_I(7)); __builtin_amdgcn_sched_barrier(0);
        }
        c_slot = c_slot + 1 == PP_DC ? 0 : c_slot + 1;
        tile_end();
        if (++c_s == nslab) { c_s = 0; ++c_r; }
    };
    // LDS reads return in order: with the four reads of blocks 4-7 still outstanding, the four in front of them (blocks 0-3 of slab g + 1)
    // are real values now -- a wait that was served half a slab ago instead of one that exposes an LDS round trip under load at the end of
    // every slab.  Blocks 4-7 are waited for before the next barrier.  This wave's VMEM queue, old -> new: .. C(g + LC - 1) | Q(g + 1) x4,
    // C(g + LC): the query fragments have landed when only the piece that follows them is outstanding.
    auto landed = [&](f32x4 (&qn)[4]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        pp_pin(ef[0], ef[1], ef[2], ef[3]);
        pp_pin(qn[0], qn[1], qn[2], qn[3]);
    };

    // ---- prologue: C(0 .. LC - 1) and slab 0's query fragments landed; slab 0's corpus fragments ------------------------------------------
    for (int i = 0; i < PP_LC; ++i) issue_c();
    issue_qregs(qA);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    [&]<int... A>(std::integer_sequence<int, A...>) { (read_slab(std::integral_constant<int, A>{}), ...); }
    (std::make_integer_sequence<int, PP_NBLK>{});
    c_slot = 1;
    landed(qA);
    // ---- main loop: two slabs per iteration (two static sets of query fragment registers) -----------------------------------------
    auto main_loop = [&](auto LAG_) __attribute__((always_inline)) {
        for (int g = 0; g < total; g += 2) {
            slab(qA, qB, LAG_);
            landed(qB);
            if (g + 1 < total) {
                slab(qB, qA, LAG_);
                landed(qA);
            }
        }
    };
#undef PP_I
    if (wv < 4) main_loop(std::false_type{});
    else main_loop(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
    dump_log();
    flush_rows();
    flush_out();
    if constexpr (DBG & 128) {  // (timing without the epilogue: the accumulators must stay live, or the compiler deletes the MFMAs with it)
        if (n_q > 1000000) {
            f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int a = 0; a < PP_NBLK; ++a) t += acc[q][qb][a];
            out[lane] = (t[0] + t[1]) + (t[2] + t[3]);
        }
    }
}

// n_q queries `first .. first + n_q - 1` of a launch_query_planes buffer over `n_queries`, each nq (<= 32) vectors, against
// a one-plane image (fp16 hi halves, or an fp16-stored corpus): out[q * out_stride + chunk], one MFMA product per multiply.
// Sixteen queries per pass over the image; n_q > 16: ceil(n_q / 16) passes in ONE launch (grid rows).
int launch_maxsim_pp(const void* image, int64_t n_rows, int32_t dim, const void* qbuf, int32_t n_queries, int32_t first, int32_t n_q,
                     int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, const uint32_t* ends_bits, float* out,
                     int64_t out_stride, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if) {
    if (nq < 1 || nq > 32 || n_q < 1 || n_q > PP_QPP * 4096 || n_rows < 1 || first < 0 || first + n_q > n_queries) return RL_ERR_UNSUPPORTED;
    if (dim % 32 || dim < 256 || !(split_scale > 0.f) || !image || !ends_bits) return RL_ERR_UNSUPPORTED;
    const int32_t nslab = dim / 32;
    const char* qfrag = static_cast<const char*>(qbuf) + (size_t)first * nslab * 4096;
    const float* qmeta = reinterpret_cast<const float*>(static_cast<const char*>(qbuf) + (size_t)n_queries * dim * 128) + 2 * (size_t)first;
    const int64_t tiles = (n_rows + PP_RT - 1) / PP_RT;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles)), (unsigned)((n_q + PP_QPP - 1) / PP_QPP)), blk(512);
#define RL_PP_LAUNCH(DBG_)                                                                                                                 \
    hipLaunchKernelGGL((maxsim_pp_kernel<DBG_, 0, false>), grid, blk, 0, s, static_cast<const char*>(image), n_rows, nslab, qfrag, qmeta, n_q, row_to_chunk, \
                       chunk_offsets, ends_bits, out, out_stride, 1.0f / split_scale, run_if, PpRows{})
#ifdef RAGLITE_EXPERIMENTS
    // Experiment builds only (libraglite_hip_exp.so, scripts/gpu_calls/): timing skeletons that skip parts of the kernel -- WRONG results.
    static const int dbg = std::getenv("RAGLITE_PP_DBG") ? std::atoi(std::getenv("RAGLITE_PP_DBG")) : 0;
    if (dbg == 2) RL_PP_LAUNCH(2);
    else if (dbg == 128) RL_PP_LAUNCH(128);
    else if (dbg == 136) RL_PP_LAUNCH(136);
    else if (dbg == 144) RL_PP_LAUNCH(144);
    else if (dbg == 160) RL_PP_LAUNCH(160);
    else if (dbg == 184) RL_PP_LAUNCH(184);
    else RL_PP_LAUNCH(0);
#else
    RL_PP_LAUNCH(0);
#endif
#undef RL_PP_LAUNCH
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- MODE 2: the candidate pass of the fused exact row top-k on the sixteen-group tile ---------------------------------------------------
// min / max |e| over every 16-row block of the image (rows past n_rows do not count): what the cosine test of the tile epilogue multiplies
__global__ __launch_bounds__(256) void block_norm_minmax_kernel(const float* __restrict__ row_norm, int64_t n_rows, int64_t n_blocks,
                                                                 float* __restrict__ out) {
    const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= n_blocks) return;
    float mn = INFINITY, mx = 0.f;
    for (int i = 0; i < 16; ++i) {
        const int64_t r = blk * 16 + i;
        if (r < n_rows) { const float v = row_norm[r]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    out[2 * blk] = mn;
    out[2 * blk + 1] = mx;
}

// The sample pass of the fused top-k on the sixteen-group tile (MODE 1): similarities of every `stride`-th 256-row tile with all nb queries,
// S[q * ld_s + 256 j + r] (ld_s = 256 x sampled tiles; rows past the corpus: -inf).  `scratch`: the query side as launch_score_planes_queries
// left it.  cosine / dot; dim % 32 == 0, dim >= 256; the row-norm array must be readable up to the next multiple of 16 rows.
int launch_pp_rows_sample(const void* image, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, const float* row_norm, int mode, float* S,
                          int64_t ld_s, int32_t stride, int n_cu, hipStream_t s, float split_scale) {
    if (nb < 1 || n_rows < 1 || dim % 32 || dim < 256 || !(split_scale > 0.f) || !image || !S || stride < 1) return RL_ERR_UNSUPPORTED;
    if (mode != SCAN_COSINE && mode != SCAN_DOT) return RL_ERR_UNSUPPORTED;
    if (mode == SCAN_COSINE && !row_norm) return RL_ERR_INVALID;
    if ((ld_s & 255) || (reinterpret_cast<uintptr_t>(S) & 15)) return RL_ERR_UNSUPPORTED;
    const int32_t nslab = dim / 32, groups = (nb + 31) / 32;
    float* frag = scratch;  // the layout of launch_score_planes_queries
    float* unscale = frag + (size_t)groups * 32 * dim;
    float* anylo = unscale + nb;
    float* qss = anylo + groups;
    const int64_t grid_cu = n_cu > 0 ? n_cu : 256;
    const int64_t T256 = (n_rows + 255) / 256, Tv = (T256 + stride - 1) / stride;
    if (Tv * 256 != ld_s) return RL_ERR_INVALID;
    PpRows rs{};
    rs.tile_begin = 0;
    rs.tile_count = (int32_t)(2 * Tv);
    rs.q_unscale = unscale; rs.q_sumsq = qss; rs.row_norm = row_norm; rs.B = nb; rs.QT = (groups + PP_QPP - 1) / PP_QPP; rs.metric = mode;
    rs.S = S; rs.ld_s = ld_s; rs.stride = stride;
    if (rs.QT > grid_cu) return RL_ERR_UNSUPPORTED;
    const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(grid_cu / rs.QT, 2 * Tv));
    hipLaunchKernelGGL((maxsim_pp_kernel<0, 1, false>), dim3((unsigned)gx, (unsigned)rs.QT), dim3(512), 0, s, static_cast<const char*>(image), n_rows,
                       nslab, reinterpret_cast<const char*>(frag), nullptr, groups, nullptr, nullptr, nullptr, nullptr, (int64_t)0,
                       1.0f / split_scale, nullptr, rs);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

size_t pp_rows_scratch_bytes(int64_t n_rows, int32_t nb, int n_cu, int32_t expected_per_query, int32_t* log_cap_out) {
    const int64_t grid = n_cu > 0 ? n_cu : 256;
    // records a wave logs: expected_per_query * nb over grid * 8 waves, x 4 for the block-wide cosine test and uneven data; at least 4096
    int64_t cap = ((int64_t)expected_per_query * nb * 4) / (grid * 8) + 1;
    cap = std::max<int64_t>(4096, (cap + 1023) & ~int64_t(1023));
    cap = std::min<int64_t>(cap, int64_t(1) << 19);  // (the record's tile index has 19 bits; 4 MiB per wave is past any sensible list anyway)
    if (log_cap_out) *log_cap_out = (int32_t)cap;
    const size_t blocks = (size_t)((n_rows + 15) / 16);
    return (size_t)grid * 8 * (size_t)cap * sizeof(uint2) + blocks * 2 * sizeof(float) + 256;
}

// The candidate pass of the fused top-k (maxsim_gemm.hip MODE 2 semantics: per-query lists of (similarity, row) for every row whose
// similarity reaches cand->tau[q]) over a ONE-PLANE image at one product per multiply, on the 128-row x 512-query tile.  `scratch`: the
// query side as launch_score_planes_queries left it; `work`: pp_rows_scratch_bytes bytes.  cosine / dot; dim % 32 == 0, dim >= 256.
int launch_pp_rows_pass(const void* image, int64_t n_rows, int32_t dim, int32_t nb, float* scratch, const float* row_norm, int mode,
                        const CandArgs* cand, void* work, int32_t log_cap, int n_cu, hipStream_t s, float split_scale, int64_t tile_begin,
                        int64_t tile_count, bool norms_ready, bool row_norm_test) {
    if (nb < 1 || n_rows < 1 || dim % 32 || dim < 256 || !(split_scale > 0.f) || !image || !cand || !work) return RL_ERR_UNSUPPORTED;
    if (mode != SCAN_COSINE && mode != SCAN_DOT) return RL_ERR_UNSUPPORTED;
    if (mode == SCAN_COSINE && !row_norm) return RL_ERR_INVALID;
    if (cand->tau_stride != 1) return RL_ERR_UNSUPPORTED;
    const int32_t nslab = dim / 32, groups = (nb + 31) / 32;
    float* frag = scratch;  // the layout of launch_score_planes_queries
    float* unscale = frag + (size_t)groups * 32 * dim;
    float* anylo = unscale + nb;
    float* qss = anylo + groups;
    const int64_t grid_cu = n_cu > 0 ? n_cu : 256;
    const int64_t Tr_all = (n_rows + PP_RT - 1) / PP_RT;
    if (tile_count < 0) tile_count = Tr_all - tile_begin;
    if (tile_begin < 0 || tile_count < 1 || tile_begin + tile_count > Tr_all) return RL_ERR_INVALID;
    const int64_t Tr = tile_count;
    PpRows rs{};
    rs.tile_begin = (int32_t)tile_begin;
    rs.tile_count = (int32_t)tile_count;
    rs.tau = cand->tau; rs.q_unscale = unscale; rs.q_sumsq = qss; rs.row_norm = row_norm; rs.B = nb; rs.QT = (groups + PP_QPP - 1) / PP_QPP;
    rs.metric = mode; rs.cand_scores = cand->scores; rs.cand_ids = cand->ids; rs.cand_cnt = cand->cnt; rs.overflow = cand->overflow; rs.cap = cand->cap;
    rs.log = static_cast<uint2*>(work);
    rs.log_cap = log_cap;
    if (rs.QT > grid_cu || Tr_all + 2 >= (int64_t(1) << 19)) return RL_ERR_UNSUPPORTED;  // one grid row per query tile; tile index bits of a record
    float* blk_mm = reinterpret_cast<float*>(static_cast<char*>(work) + (size_t)grid_cu * 8 * (size_t)log_cap * sizeof(uint2));
    rs.blk_minmax = blk_mm;
    if (mode == SCAN_COSINE && !norms_ready) {  // (once per search: the rounds of a search share it)
        const int64_t n_blocks = (n_rows + 15) / 16;
        hipLaunchKernelGGL(block_norm_minmax_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, s, row_norm, n_rows, n_blocks, blk_mm);
    }
    // blockIdx.y = the query tile; the workgroups of a grid row share the row tiles (QT <= CUs: checked above)
    const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(grid_cu / rs.QT, Tr));
    const dim3 grid((unsigned)gx, (unsigned)rs.QT), blk(512);
#define RL_PP_ROWS(DBG_, RN_)                                                                                                             \
    hipLaunchKernelGGL((maxsim_pp_kernel<DBG_, 2, RN_>), grid, blk, 0, s, static_cast<const char*>(image), n_rows, nslab, reinterpret_cast<const char*>(frag), \
                       nullptr, groups, nullptr, nullptr, nullptr, nullptr, (int64_t)0, 1.0f / split_scale, nullptr, rs)
#ifdef RAGLITE_EXPERIMENTS  // timing skeleton (wrong results): 128 = no block epilogues
    static const int dbg = std::getenv("RAGLITE_PP_ROWS_DBG") ? std::atoi(std::getenv("RAGLITE_PP_ROWS_DBG")) : 0;
    if (dbg == 128) RL_PP_ROWS(128, false);
    else if (row_norm_test) RL_PP_ROWS(0, true);
    else RL_PP_ROWS(0, false);
#else
    // (ROWNORM: the row-by-row cosine test in the hit path, for indexes whose row norms differ wildly)
    if (row_norm_test) RL_PP_ROWS(0, true);
    else RL_PP_ROWS(0, false);
#endif
#undef RL_PP_ROWS
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// ---- the matrix pipe's SUSTAINED rate, measured next to the pass kernel (rl_time_kernel kind 9; bench.py: roofline.sustained) -----------
// The pass kernel's MFMA stream and nothing else: 8 waves per workgroup (two per SIMD), one workgroup per CU, 128 accumulator registers per
// wave, 32 v_mfma_f32_16x16x32_f16 per iteration (8 "corpus" fragments x 4 "query" fragments, like a K slab of the pass), operands that
// differ from lane to lane and from fragment to fragment (hash of the lane: the multipliers toggle like real data, not like constants).
// No loads, no LDS, no epilogue: what the chip delivers when only the matrix pipe is asked -- at the clock it settles at under that load
// (the nominal 2.5 PFLOP/s is 16 cycles per MFMA at 2.4 GHz; under a full MFMA load the shader clock settles near 1.75-1.85 GHz).
__global__ __launch_bounds__(512, 2) void mfma_f16_rate_kernel(float* __restrict__ out, int32_t iters) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    f32x4 ef[8], qf[4];
    auto frag = [&](uint32_t tag) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t h = (lane * 0x9E3779B1u) ^ ((tag * 4u + (uint32_t)i + 1u) * 0x85EBCA77u) ^ (wv * 0xC2B2AE3Du) ^ (blockIdx.x * 0x27D4EB2Fu);
            h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
            // two fp16 values in [-1, 1): sign, exponent 0b01110 or 0b01101 or below (values < 1), random mantissa
            const uint32_t lo = (h & 0x83FFu) | (((h >> 10) & 3u) + 11u) << 10, hi = ((h >> 16) & 0x83FFu) | (((h >> 26) & 3u) + 11u) << 10;
            v[i] = __uint_as_float(lo | (hi << 16));
        }
        return v;
    };
#pragma unroll
    for (int a = 0; a < 8; ++a) ef[a] = frag(a);
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = frag(8 + c);
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int32_t t = 0; t < iters; ++t) {
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(ef[a]), pp_h(qf[c]), acc[a][c], 0, 0, 0);
    }
    f32x4 t4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) t4 += acc[a][c];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = (t4[0] + t4[1]) + (t4[2] + t4[3]);
}

// One launch: n_cu workgroups x 8 waves x iters x 32 MFMAs of 2 * 16 * 16 * 32 flop; `out`: n_cu * 512 floats.  *flops = what the launch executes.
int launch_mfma_f16_rate(float* out, int n_cu, int32_t iters, hipStream_t s, double* flops) {
    if (!out || iters < 1) return RL_ERR_INVALID;
    const int grid = n_cu > 0 ? n_cu : 256;
    hipLaunchKernelGGL(mfma_f16_rate_kernel, dim3((unsigned)grid), dim3(512), 0, s, out, iters);
    RL_HIP(hipGetLastError());
    if (flops) *flops = (double)grid * 8.0 * (double)iters * 32.0 * (2.0 * 16 * 16 * 32);
    return RL_OK;
}

}  // namespace rl
