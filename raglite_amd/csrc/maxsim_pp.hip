// a9 at batch scale, the approximate pass of the headline pipeline (api.hip: rl_maxsim_topk_batch): MaxSim of SIXTEEN queries
// (up to 32 vectors each) per corpus pass over the HI image -- the fp16 hi halves of the corpus in the one-plane image layout of
// maxsim_gemm.hip (preformat / presplit_hi_rows_kernel) -- at ONE fp16 MFMA product per multiply (q_hi . e_hi):
//
//   out[q * out_stride + c] ~ sum_{i < nq} max_{j in chunk c} Q_q[i] . D[j]      q = 0 .. n_q - 1  (n_q <= 16)
//
// Multi-query generalisation of src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174 behind the reranker plugin
// call src/raglite/_search.py:394-396; what it drops (the corpus' and the queries' lo halves) is bounded rigorously by the caller,
// which re-scores every chunk the bound cannot rule out with exact fp32 products (maxsim_pairs_kernel).
//
// Why a new kernel next to maxsim_gemm_kernel<.., HALF>.  Measured in round 3 (profiles/r03_a_*): that kernel's one-product pass
// takes 0.71 ms per eight queries where its 32 MFMAs per wave and slab are 0.26 ms of matrix pipe; a six-slot ring with the query
// fragments three slabs ahead (its HO variant) gets 0.65 -- the corpus stream was not what it waited for.  A wave there reads its
// corpus fragments from LDS ONE pair of blocks ahead: with three products a pair covers 12 MFMAs (~400 cycles for the two waves of
// a SIMD), with one product 4 (~130), less than an LDS round trip under load -- the matrix pipe idles on lgkmcnt in every pair of
// every slab.  And at eight queries per pass the pass needs 2 GB / 0.3 ms = 6.7 TB/s of HBM once the matrix pipe is fed.  So:
//   * tile = 128 corpus rows x 512 query vectors (16 queries), K slabs of 32; wave w owns queries 2w and 2w + 1: the same 128
//     accumulator registers (2 queries x 2 blocks of 16 vectors x 8 blocks of 16 rows), half the corpus bytes per query -- HBM
//     3.3 TB/s at full matrix rate -- and 12 fragment reads per 32 MFMAs instead of 18;
//   * EVERY operand goes through LDS by `global_load_lds_dwordx4`: corpus slabs (8 KiB, `nt`) in a ring of 6, query slabs (the 16
//     queries' hi fragments, 32 KiB, L2-resident) in a ring of 3 -- no VMEM result is ever waited for in registers, so no wait
//     drains the look-ahead (VMEM retires in order).  Waves 0 and 4 feed the corpus ring and nothing else: their `s_waitcnt vmcnt`
//     leaves four slabs (32 KiB per CU) of HBM reads in flight; the other six feed the query ring (5, 5, 6 pieces per slab);
//     (a wave issues its pieces of slab g + D during the first MFMA groups of slab g, whose slot the barrier just freed);
//   * a wave's fragments of slab g + 1 are read from LDS WHILE it multiplies slab g from registers -- each corpus fragment
//     register is re-loaded right after its four MFMAs, the query fragments alternate between two register sets -- i.e. a full
//     slab (32 MFMAs, >= 512 cycles) ahead: the MFMA stream never waits for LDS;
//   * ONE workgroup barrier per slab, HALF-WAY through it: it certifies slab g + 1 as landed (every feeder waited for its pieces)
//     and slab g's LDS slot as free (everybody read it during slab g - 1); the four MFMA groups in front of it depend on nobody, so
//     a late feeder or a slow wave costs matrix-pipe time only when it is later than that;
//   * tile epilogue as in maxsim_gemm_kernel (DPP segmented max-scan along the 16 rows of a block, sum over the 32 query vectors,
//     store by the chunk's end row's lane), once per query of the wave.  Its stores share the in-order VMEM counter with the
//     DMAs: the feeders count them (wave-uniform) and widen their next waits by exactly that many.
// Tried on top of this and removed (profiles/r03_j, r03_k; commit 'maxsim_pp2_kernel ... for the record'): eight queries per pass
// over TWO row streams per workgroup (32 KiB per slab instead of 40) with the two waves of a SIMD in alternating phases (one multiplies,
// its partner loads and feeds).  Correct -- bit-identical -- and without its epilogue it sits on the operand stream with the MFMAs
// hidden (0.36 ms per eight queries); but the tile epilogue (2 200 straight-line VALU / SALU instructions per wave, ~10 cycles each
// when a wave runs alone on its SIMD) then runs once per GROUP, back to back: 0.72 ms per eight queries against 0.58 here.  The
// epilogue, not the main loop, is what the next version has to make cheaper.
// Deterministic (fixed MFMA order per (query vector, row), fixed scan / sum order; independent of the grid); integer-valued
// data is exact.  Needs an index without empty chunks (a chunk is found by counting chunk ends), nq <= 32, dim % 32 == 0.
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
constexpr int PP_WAVES = 8;
constexpr int PP_CSLOT = PP_NBLK * 1024;         // corpus slab: 8 blocks x 1 KiB
constexpr int PP_QSLOT = PP_QPP * 2 * 1024;      // query slab: 16 queries x 2 blocks of 16 vectors x 1 KiB

__device__ __forceinline__ int64_t pp_uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// One 1-KiB piece HBM / L2 -> LDS: lane i copies 16 B from src + 16 i to lds + 16 i.  Invisible to the compiler's vmcnt
// bookkeeping (asynchronous; certified by the explicit waits below).  NT: streamed once (corpus); plain: re-read by every
// workgroup from L2 (queries).
template <bool NT>
__device__ __forceinline__ void pp_dma(uint32_t lds, const char* src, uint32_t lane16) {
    const uint32_t l = __builtin_amdgcn_readfirstlane(lds);
    const char* const p = reinterpret_cast<const char*>(pp_uniform_i64(reinterpret_cast<int64_t>(src)));
    if constexpr (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(l), "v"(lane16), "s"(p) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane16), "s"(p) : "memory", "m0");
}

// s_waitcnt vmcnt(BASE + extra), extra = 0 .. 16 wave-uniform (the immediate cannot come from a register)
template <int BASE>
__device__ __forceinline__ void pp_wait_vm(int extra) {
    if (extra == 0) {  // every slab but the few after a tile's epilogue
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE) : "memory");
        return;
    }
#define PP_CASE(I) case I: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BASE + I) : "memory"); break;
    switch (extra) {
        PP_CASE(1) PP_CASE(2) PP_CASE(3) PP_CASE(4) PP_CASE(5) PP_CASE(6) PP_CASE(7) PP_CASE(8)
        PP_CASE(9) PP_CASE(10) PP_CASE(11) PP_CASE(12) PP_CASE(13) PP_CASE(14) PP_CASE(15) PP_CASE(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // (never: at most 16 stores per tile and wave)
    }
#undef PP_CASE
}
}  // namespace

#define PP_MAX_DPP(dst, src, x, CTRLSTR) asm volatile("v_max_f32_dpp %0, %1, %2 " CTRLSTR " row_mask:0xf bank_mask:0xf" : "=&v"(dst) : "v"(src), "v"(x))
#define PP_SELECT(x, t, mask) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(t), "s"(mask))

// One LDS fragment read (64 lanes x 16 B) into 4 VGPRs, invisible to the compiler's lgkmcnt bookkeeping -- its own waits would sit in
// front of every MFMA group (the loop body is not one basic block, and the pass gives up at the joins): the value is valid after the
// `s_waitcnt lgkmcnt(0)` at the end of the slab + pp_pin().
template <int OFF>
__device__ __forceinline__ void pp_read(f32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pp_pin(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ h16x8 pp_h(const f32x4& v) {
    h16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// DBG (timing experiments only, RAGLITE_PP_DBG; wrong results): 1 = no scan steps in the epilogue, 2 = no MFMAs, 8 = no LDS fragment reads,
// 16 = no corpus DMAs, 32 = no query DMAs
// PP_DC / PP_DQ: ring depths in slabs (corpus: 8 KiB each, query: 32 KiB each; at most 160 KiB together).
// FEED 0: waves 0, 4 feed the corpus ring, the other six the query ring.  FEED 1: waves 0-3 (one per SIMD) feed both rings -- per slab 8
// query pieces, then 2 corpus pieces -- and waves 4-7 only multiply: a VMEM instruction blocks its wave until the address path takes it
// (~35 cycles per KiB with every CU streaming), and a blocked wave issues no MFMAs; with one feeder per SIMD its partner keeps the
// matrix pipe busy meanwhile.
template <int DBG, int PP_DC, int PP_DQ, int FEED>
__global__ __launch_bounds__(512, 2) void maxsim_pp_kernel(const char* __restrict__ planes, int64_t n_rows, int32_t nslab,
                                                           const char* __restrict__ qfrag, const float* __restrict__ qmeta, int32_t n_q,
                                                           const int32_t* __restrict__ row_to_chunk, const int64_t* __restrict__ chunk_offsets,
                                                           const uint32_t* __restrict__ ends_bits, float* __restrict__ out, int64_t out_stride,
                                                           float inv_e_scale, const uint32_t* __restrict__ run_if, unsigned long long* __restrict__ trace) {
    constexpr int PP_QOFF = PP_DC * PP_CSLOT, PP_LDS = PP_QOFF + PP_DQ * PP_QSLOT;
    static_assert(PP_LDS <= 160 * 1024 && PP_DC >= 3 && PP_DQ >= 3, "ring depths");
    static_assert(FEED == 0 || (PP_DC == 4 && PP_DQ == 3), "FEED 1: query slab g + 3 and corpus slab g + 4 are issued during slab g");
    // DBG & 64 (RAGLITE_PP_TRACE=1): s_memtime stamps of workgroup 7, slabs 64..79, in LDS behind the rings, copied out at the end:
    // [slab - 64][wave][0: top of the slab, 1: after the first four MFMA groups, 2: after the feeder's wait, 3: end of the slab]
    constexpr bool TRACE = (DBG & 64) != 0;
    __shared__ __attribute__((aligned(16))) char smem[PP_LDS + (TRACE ? 16 * 8 * 4 * 8 : 0)];
    [[maybe_unused]] int g_now = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (blockIdx.x == 7 && g_now >= 64 && g_now < 80 && (threadIdx.x & 63) == 0)
                reinterpret_cast<unsigned long long*>(smem + PP_LDS)[((g_now - 64) * 8 + (threadIdx.x >> 6)) * 4 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    if (run_if && __builtin_amdgcn_readfirstlane((int)*run_if) == 0) return;  // whole grid: a guarded launch that is not needed
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();
    const int64_t G = gridDim.x, b = blockIdx.x;
    // Chunk-aligned row range of this workgroup (as in maxsim_gemm.hip): first chunk boundary at or after n_rows * b / G.
    auto boundary = [&](int64_t t) -> int64_t {
        if (t <= 0) return 0;
        if (t >= n_rows) return n_rows;
        const int32_t c = row_to_chunk[t];
        const int64_t c0 = chunk_offsets[c], c1 = chunk_offsets[c + 1];
        return c0 == t ? t : c1;
    };
    const int32_t r_lo = (int32_t)pp_uniform_i64(boundary((n_rows * b) / G));
    const int32_t r_hi = (int32_t)pp_uniform_i64((b + 1 == G) ? n_rows : boundary((n_rows * (b + 1)) / G));
    if (r_hi <= r_lo) return;  // whole workgroup
    const int32_t org = r_lo & ~15;  // tiles start on a 16-row block of the image
    const int nt = (r_hi - org + PP_RT - 1) / PP_RT;
    const int total = nt * nslab;  // K slabs this workgroup consumes, tile after tile
    const int32_t last_blk = (int32_t)((n_rows + 15) >> 4) - 1;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;
    const int fj = lane & 15;

    // ---- this wave's queries ------------------------------------------------------------------------------------------------
    const bool has0 = 2 * wv < n_q, has1 = 2 * wv + 1 < n_q;  // wave-uniform
    const float unscale0 = has0 ? qmeta[2 * (2 * wv)] * inv_e_scale : 0.f, unscale1 = has1 ? qmeta[2 * (2 * wv + 1)] * inv_e_scale : 0.f;
    float* const out0 = out + (int64_t)(has0 ? 2 * wv : 0) * out_stride;
    float* const out1 = out + (int64_t)(has1 ? 2 * wv + 1 : 0) * out_stride;

    // ---- feeder duty -----------------------------------------------------------------------------------------------------------
    // waves 0, 4: corpus blocks 4 (wv >> 2) .. + 3 of every slab, DC slabs ahead; waves 1-3, 5-7: query pieces
    // 16 (wv >> 2) + j0 .. + np - 1 of every slab (piece p = 2 * query + block of 16 vectors), DQ slabs ahead.
    const bool cfeed = (wv & 3) == 0;
    const int grp = wv >> 2, wi = wv & 3;
    const int j0 = cfeed ? 0 : (wi - 1) * 5, np = cfeed ? 4 : (wi == 3 ? 6 : 5);
    const char* fbase[6];  // piece bases at slab 0 (corpus: of the tile being fetched)
    int f_tile = 0, f_s = 0, f_slot = 0;  // position of the NEXT slab this wave fetches
    auto feed_tile = [&](int t) __attribute__((always_inline)) {
        const int32_t b0 = ((org + t * PP_RT) >> 4) + 4 * grp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int32_t blk = b0 + i;
            blk = blk < last_blk ? blk : last_blk;  // past the image: harmless re-read of the last block, never emitted
            fbase[i] = planes + pp_uniform_i64((int64_t)blk * nslab * 1024);
        }
    };
    if (cfeed) {
        feed_tile(0);
        fbase[4] = fbase[5] = fbase[0];
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = 16 * grp + j0 + (i < np ? i : 0);
            int ql = p >> 1;
            ql = ql < n_q ? ql : n_q - 1;  // (clamped: pieces of queries the pass does not have are copied from a valid one, never used)
            fbase[i] = qfrag + pp_uniform_i64(((int64_t)ql * nslab * 4 + 2 * (p & 1)) * 1024);  // hi fragments of block p & 1
        }
    }
    // Piece i of the NEXT slab this wave fetches (i < np; wave-uniform), then issue_advance() once per slab.
    auto issue_piece = [&](int i) __attribute__((always_inline)) {
        if (i >= np) return;
        if constexpr (DBG & 16) { if (cfeed) return; }
        if constexpr (DBG & 32) { if (!cfeed) return; }
        if (cfeed) pp_dma<true>(lds_base + (uint32_t)(f_slot * PP_CSLOT + (4 * grp + i) * 1024), fbase[i] + (int64_t)f_s * 1024, lane16);
        else pp_dma<false>(lds_base + (uint32_t)(PP_QOFF + f_slot * PP_QSLOT + (16 * grp + j0 + i) * 1024), fbase[i] + (int64_t)f_s * 4096, lane16);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        if (++f_s == nslab) {  // corpus: past the end re-fetch the last tile (keeps the vmcnt bookkeeping uniform)
            f_s = 0;
            if (cfeed && f_tile + 1 < nt) { ++f_tile; feed_tile(f_tile); }
        }
        f_slot = f_slot + 1 == (cfeed ? PP_DC : PP_DQ) ? 0 : f_slot + 1;
    };
    auto issue_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(i);
        issue_advance();
    };
    // ---- FEED 1: waves 0-3 fetch query pieces 8 wv .. + 7 of slab g + 3, then corpus blocks 2 wv, 2 wv + 1 of slab g + 4 ----------
    const bool feeder1 = wv < 4;
    const char* qb1[8];
    const char* cb1[2];
    int fq_s = 0, fq_slot = 0, fc_s = 0, fc_tile = 0, fc_slot = 0;
    auto feed1_tile = [&](int t) __attribute__((always_inline)) {
        const int32_t b0 = ((org + t * PP_RT) >> 4) + 2 * (wv & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int32_t blk = b0 + i;
            blk = blk < last_blk ? blk : last_blk;
            cb1[i] = planes + pp_uniform_i64((int64_t)blk * nslab * 1024);
        }
    };
    if constexpr (FEED == 1) {
        feed1_tile(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = 8 * (wv & 3) + i;
            int ql = p >> 1;
            ql = ql < n_q ? ql : n_q - 1;
            qb1[i] = qfrag + pp_uniform_i64(((int64_t)ql * nslab * 4 + 2 * (p & 1)) * 1024);
        }
    }
    auto issue1_q = [&](int i) __attribute__((always_inline)) {
        if (!feeder1) return;
        pp_dma<false>(lds_base + (uint32_t)(PP_QOFF + fq_slot * PP_QSLOT + (8 * wv + i) * 1024), qb1[i] + (int64_t)fq_s * 4096, lane16);
    };
    auto advance1_q = [&]() __attribute__((always_inline)) {
        if (++fq_s == nslab) fq_s = 0;
        fq_slot = fq_slot + 1 == PP_DQ ? 0 : fq_slot + 1;
    };
    auto issue1_c = [&]() __attribute__((always_inline)) {
        if (!feeder1) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) pp_dma<true>(lds_base + (uint32_t)(fc_slot * PP_CSLOT + (2 * wv + i) * 1024), cb1[i] + (int64_t)fc_s * 1024, lane16);
    };
    auto advance1_c = [&]() __attribute__((always_inline)) {
        if (++fc_s == nslab) {
            fc_s = 0;
            if (fc_tile + 1 < nt) { ++fc_tile; feed1_tile(fc_tile); }
        }
        fc_slot = fc_slot + 1 == PP_DC ? 0 : fc_slot + 1;
    };
    // Wait until this wave's pieces of every slab but the newest `keep` have landed (+ `extra` newer stores).
    int st_pending = 0, st_slabs = 0;  // epilogue stores newer than DMAs this wave still has to certify: how many, for how many more slabs
    auto certify = [&]() __attribute__((always_inline)) {
        const int extra = st_slabs > 0 ? st_pending : 0;
        if constexpr (FEED == 1) {
            // queue, old -> new, at the end of slab g: .. Q(g+2) x8 | C(g+3) x2, Q(g+3) x8, C(g+4) x2: slab g + 2 is complete when 12 are left
            if (feeder1) pp_wait_vm<12>(extra);
        } else {
            if (cfeed) pp_wait_vm<(PP_DC - 2) * 4>(extra);
            else if (np == 6) pp_wait_vm<(PP_DQ - 2) * 6>(extra);
            else pp_wait_vm<(PP_DQ - 2) * 5>(extra);
        }
        if (st_slabs > 0) --st_slabs;
    };

    // ---- accumulators: S^T[query vector 16 qb + 4 g + u][corpus row 16 a + j], lane = 16 g + j; [query of the wave][qb][a] ------
    f32x4 acc[2][2][PP_NBLK];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int a = 0; a < PP_NBLK; ++a) acc[q][qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float carry[2][8];  // scanned maxima of the previous block (lane 15 of each DPP row = the chunk still open)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r) carry[q][r] = -INFINITY;
    uint32_t prev_last_end = 1;  // is the row before the tile's first row the last row of its chunk?
    // chunk ordinal of the next block's first row: no chunk is empty, so it advances by the number of chunk ends in a block -- one scalar
    // load per workgroup instead of one (dependent, ~250 cycles each) per block
    int32_t ord_run = __builtin_amdgcn_readfirstlane(row_to_chunk[org]);

    // ---- fragment registers ---------------------------------------------------------------------------------------------------
    f32x4 ef[PP_NBLK];      // corpus blocks of the slab being multiplied (each re-loaded for the next slab right after its MFMAs)
    f32x4 qA[4], qB[4];     // query fragments [2 * query of the wave + qb], two sets alternating over slabs
    int c_slot = 0, q_slot = 0;  // ring slots of the slab whose fragments are READ next
    const uint32_t rd_c = lds_base + lane16, rd_q = lds_base + (uint32_t)(PP_QOFF + 4 * wv * 1024) + lane16;
    auto read_slab = [&](f32x4 (&qn)[4], auto A_) __attribute__((always_inline)) {  // reads of step a: corpus block a (+ query fragment a)
        constexpr int a = decltype(A_)::value;
        if constexpr (!(DBG & 8)) {
            pp_read<a * 1024>(ef[a], rd_c + (uint32_t)(c_slot * PP_CSLOT));
            if constexpr (a < 4) pp_read<a * 1024>(qn[a], rd_q + (uint32_t)(q_slot * PP_QSLOT));
        }
    };

    // ---- one K slab: MFMAs of slab g from registers; HALF-WAY through, the feeders' wait and the workgroup barrier -- they certify slab
    // g + 1 as landed and slab g's LDS slot as free -- then, between the remaining MFMAs, the fragment reads of slab g + 1 and this
    // wave's DMAs of slab g + D.  The first four MFMA groups need nothing from anybody: whoever is late (a feeder waiting for its
    // pieces, the younger wave of a SIMD) is late while the matrix pipe still has work.  No compiler-visible memory operation in
    // here: the body carries no wait but the two written out.
    auto mfma_group = [&](f32x4 (&q)[4], auto A_) __attribute__((always_inline)) {
        constexpr int A = decltype(A_)::value;
        if constexpr (!(DBG & 2)) {
            acc[0][0][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(q[0]), pp_h(ef[A]), acc[0][0][A], 0, 0, 0);
            acc[0][1][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(q[1]), pp_h(ef[A]), acc[0][1][A], 0, 0, 0);
            acc[1][0][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(q[2]), pp_h(ef[A]), acc[1][0][A], 0, 0, 0);
            acc[1][1][A] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pp_h(q[3]), pp_h(ef[A]), acc[1][1][A], 0, 0, 0);
        }
    };
    auto slab = [&](f32x4 (&q)[4], f32x4 (&qn)[4]) __attribute__((always_inline)) {
        [&]<int... A>(std::integer_sequence<int, A...>) {
            ((mfma_group(q, std::integral_constant<int, A>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, 4>{});
        stamp(1);
        certify();  // this wave's pieces of slab g + 1
        stamp(2);
        asm volatile("s_barrier" ::: "memory");  // slab g + 1 is readable; everybody finished reading slab g (lgkmcnt(0) at the end of slab g - 1)
        [&]<int... A>(std::integer_sequence<int, A...>) { (read_slab(qn, std::integral_constant<int, A>{}), ...); }
        (std::make_integer_sequence<int, 4>{});  // corpus blocks 0-3 (their MFMAs are done) and the four query fragments of slab g + 1
        [&]<int... B>(std::integer_sequence<int, B...>) {
            (([&] {
                 constexpr int A = 4 + B;
                 mfma_group(q, std::integral_constant<int, A>{});
                 read_slab(qn, std::integral_constant<int, A>{});
                 // this wave's DMAs of slab g + D, spread over the steps: in flight while it multiplies
                 if constexpr (FEED == 1) {
                     issue1_q(2 * B);
                     issue1_q(2 * B + 1);
                 } else {
                     issue_piece(B);
                     if constexpr (B < 2) issue_piece(4 + B);
                 }
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, 4>{});
        if constexpr (FEED == 1) {
            issue1_c();
            advance1_q();
            advance1_c();
        } else {
            issue_advance();
        }
        c_slot = c_slot + 1 == PP_DC ? 0 : c_slot + 1;
        q_slot = q_slot + 1 == PP_DQ ? 0 : q_slot + 1;
    };
    auto landed = [&](f32x4 (&qn)[4]) __attribute__((always_inline)) {  // after lgkmcnt(0): the fragments just read are real values now
        pp_pin(ef[0], ef[1], ef[2], ef[3]);
        pp_pin(ef[4], ef[5], ef[6], ef[7]);
        pp_pin(qn[0], qn[1], qn[2], qn[3]);
    };

    // ---- tile epilogue: per-chunk maxima along the DPP rows, sum over the query vectors, store ------------------------------------
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int32_t row0 = org + t * PP_RT;
        // "last row of its chunk" bits of the tile's 128 rows: 5 words from row0 / 32, shifted by 16 when row0 is odd in blocks
        uint32_t m[5];
        const uint32_t* const eb = ends_bits + (row0 >> 5);
#pragma unroll
        for (int i = 0; i < 5; ++i) m[i] = eb[i];
        const bool odd = (row0 & 16) != 0;
        uint32_t mm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) mm[i] = odd ? (m[i] >> 16) | (m[i + 1] << 16) : m[i];
        int n_st = 0;
        if (has0) {
#pragma unroll
            for (int a = 0; a < PP_NBLK; ++a) {
                const uint32_t E = (mm[a >> 1] >> (16 * (a & 1))) & 0xffffu;
                const int32_t base = row0 + 16 * a;
                const int32_t ord0 = ord_run;
                ord_run += __builtin_popcount(E);
                // lanes whose shifted neighbour belongs to the same chunk: no chunk end in rows [j - d, j - 1]
                const uint32_t O1 = E << 1, O2 = O1 | (O1 << 1), O4 = O2 | (O2 << 2), O8 = O4 | (O4 << 4);
                const uint64_t rep = 0x0001000100010001ull;
                const uint64_t F1 = (uint64_t)(~O1 & 0xfffeu) * rep, F2 = (uint64_t)(~O2 & 0xfffcu) * rep;
                const uint64_t F4 = (uint64_t)(~O4 & 0xfff0u) * rep, F8 = (uint64_t)(~O8 & 0xff00u) * rep;
                const uint64_t C0 = prev_last_end ? 0ull : rep;  // row 0 continues the chunk open at the end of the previous block
                int32_t lo = r_lo - base, hi = r_hi - base;  // rows of this block inside the workgroup's range
                lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
                hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                const uint32_t EM = E & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                const uint32_t below = E & ((1u << fj) - 1u);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && !has1) continue;  // wave-uniform
                    float x[8], tmp[8];
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[4 * qb + u] = acc[q][qb][a][u];
#define PP_STEP(SRC, CTRLSTR, MASK)                                                    \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) PP_MAX_DPP(tmp[r], SRC, x[r], CTRLSTR); \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) PP_SELECT(x[r], tmp[r], MASK);
                    if constexpr (!(DBG & 1)) {
                        PP_STEP(carry[q][r], "row_ror:1", C0)  // lane 0 <- lane 15 of the previous block's scan
                        PP_STEP(x[r], "row_shr:1", F1)
                        PP_STEP(x[r], "row_shr:2", F2)
                        PP_STEP(x[r], "row_shr:4", F4)
                        PP_STEP(x[r], "row_shr:8", F8)
                    }
#undef PP_STEP
#pragma unroll
                    for (int r = 0; r < 8; ++r) carry[q][r] = x[r];
                    if (EM != 0u) {  // wave-uniform: some chunk of this workgroup ends in this block
                        float tsum = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                        // + the other three lane groups (query vectors 4 g .. 4 g + 3 live in group g): v_permlane16/32_swap of a register
                        // with itself pairs lane l with l ^ 16 / l ^ 32 -- the same two operands per add as a shuffle, no LDS round trip
                        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(tsum), __float_as_uint(tsum), false, false);
                        tsum = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
                        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(tsum), __float_as_uint(tsum), false, false);
                        tsum = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
                        if (lane < 16 && ((EM >> lane) & 1u)) (q == 0 ? out0 : out1)[ord0 + __builtin_popcount(below)] = tsum * (q == 0 ? unscale0 : unscale1);
                        ++n_st;
                    }
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) acc[q][qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                prev_last_end = (E >> 15) & 1u;
            }
        } else {
            prev_last_end = (mm[3] >> 31) & 1u;
            ord_run += __builtin_popcount(mm[0]) + __builtin_popcount(mm[1]) + __builtin_popcount(mm[2]) + __builtin_popcount(mm[3]);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int a = 0; a < PP_NBLK; ++a) acc[q][qb][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // The stores are newer than every DMA issued so far (up to slab g + D): they count in the waits that certify slabs
        // g + 2 .. g + D, i.e. in this slab's wait and the next D - 2.
        st_pending = n_st;
        st_slabs = n_st > 0 ? (FEED == 1 ? 2 : (cfeed ? PP_DC - 1 : PP_DQ - 1)) : 0;
    };

    // ---- prologue: slabs 0 .. D - 1 of this wave's rings in flight; slab 0 landed -> its fragments into registers ---------------------
    if constexpr (FEED == 1) {
        if (feeder1) {  // Q(0..2), C(0..3): the steady state issues Q(g + 3), C(g + 4) during slab g
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) issue1_q(j);
                advance1_q();
            }
            for (int i = 0; i < 4; ++i) { issue1_c(); advance1_c(); }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        const int D = cfeed ? PP_DC : PP_DQ;
        for (int i = 0; i < D; ++i) issue_slab();
        if (cfeed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PP_DC - 1) * 4) : "memory");
        else if (np == 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PP_DQ - 1) * 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PP_DQ - 1) * 5) : "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    [&]<int... A>(std::integer_sequence<int, A...>) { (read_slab(qA, std::integral_constant<int, A>{}), ...); }
    (std::make_integer_sequence<int, PP_NBLK>{});
    c_slot = 1;
    q_slot = 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    landed(qA);
    // ---- main loop: two slabs per iteration (two static sets of query fragment registers) -----------------------------------------
    int c_s = 0, c_tile = 0;
    auto step = [&](f32x4 (&q)[4], f32x4 (&qn)[4]) __attribute__((always_inline)) {
        stamp(0);
        slab(q, qn);
        if (++c_s == nslab) {
            epilogue(c_tile);
            c_s = 0;
            ++c_tile;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragments of slab g + 1 are in its registers
        landed(qn);
        stamp(3);
        if constexpr (TRACE) ++g_now;
    };
    for (int g = 0; g < total; g += 2) {
        step(qA, qB);
        if (g + 1 < total) step(qB, qA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
    if constexpr (TRACE) {
        __syncthreads();
        if (blockIdx.x == 7 && trace)
            for (int i = threadIdx.x; i < 16 * 8 * 4; i += blockDim.x) trace[i] = reinterpret_cast<unsigned long long*>(smem + PP_LDS)[i];
    }
}

// n_q (1..16) queries `first .. first + n_q - 1` of a launch_query_planes buffer over `n_queries`, each nq (<= 32) vectors, against
// a one-plane image (fp16 hi halves, or an fp16-stored corpus): out[q * out_stride + chunk], one MFMA product per multiply.
int launch_maxsim_pp(const void* image, int64_t n_rows, int32_t dim, const void* qbuf, int32_t n_queries, int32_t first, int32_t n_q,
                     int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, const uint32_t* ends_bits, float* out,
                     int64_t out_stride, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if) {
    if (nq < 1 || nq > 32 || n_q < 1 || n_q > PP_QPP || n_rows < 1 || first < 0 || first + n_q > n_queries) return RL_ERR_UNSUPPORTED;
    // dim >= 256: a tile's epilogue stores must have left the VMEM counter's window before the next tile's (see certify())
    if (dim % 32 || dim < 256 || !(split_scale > 0.f) || !image || !ends_bits) return RL_ERR_UNSUPPORTED;
    const int32_t nslab = dim / 32;
    const char* qfrag = static_cast<const char*>(qbuf) + (size_t)first * nslab * 4096;
    const float* qmeta = reinterpret_cast<const float*>(static_cast<const char*>(qbuf) + (size_t)n_queries * dim * 128) + 2 * (size_t)first;
    const int64_t tiles = (n_rows + PP_RT - 1) / PP_RT;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles))), blk(512);
    static const int dbg = std::getenv("RAGLITE_PP_DBG") ? std::atoi(std::getenv("RAGLITE_PP_DBG")) : 0;  // timing experiments only
#define RL_PP_LAUNCH(DBG_, DC_, DQ_, FEED_)                                                                                               \
    hipLaunchKernelGGL((maxsim_pp_kernel<DBG_, DC_, DQ_, FEED_>), grid, blk, 0, s, static_cast<const char*>(image), n_rows, nslab, qfrag, qmeta, \
                       n_q, row_to_chunk, chunk_offsets, ends_bits, out, out_stride, 1.0f / split_scale, run_if, trace)
    static unsigned long long* trace = [] {
        unsigned long long* p = nullptr;
        if (std::getenv("RAGLITE_PP_TRACE")) { (void)hipMalloc(&p, 16 * 8 * 4 * 8); (void)hipMemset(p, 0, 16 * 8 * 4 * 8); }
        return p;
    }();
    static const int feed = std::getenv("RAGLITE_PP_FEED") ? std::atoi(std::getenv("RAGLITE_PP_FEED")) : 1;  // A/B: 0 = every wave feeds (1.31 ms per pass against 1.155, profiles/r03_h)
    if (trace && n_q == PP_QPP) {  // diagnostic build: dump the 10th launch's slab timeline to stderr
        static int calls = 0;
        if (feed == 1) RL_PP_LAUNCH(64, 4, 3, 1);
        else RL_PP_LAUNCH(64, 6, 3, 0);
        if (++calls == 10) {
            static unsigned long long h[16 * 8 * 4];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
            fprintf(stderr, "PPTRACE columns: top-of-slab after-four-MFMA-groups after-feeder-wait end-of-slab (shader cycles, relative)\n");
            for (int g = 0; g < 16; ++g)
                for (int w = 0; w < 8; ++w) {
                    fprintf(stderr, "PPTRACE slab %d wave %d:", g + 64, w);
                    for (int k = 0; k < 4; ++k) fprintf(stderr, " %7lld", (long long)(h[(g * 8 + w) * 4 + k] - h[0]));
                    fprintf(stderr, "\n");
                }
        }
        RL_HIP(hipGetLastError());
        return RL_OK;
    }
    if (feed == 1) {
        if (dbg == 11) RL_PP_LAUNCH(11, 4, 3, 1);
        else if (dbg == 2) RL_PP_LAUNCH(2, 4, 3, 1);
        else RL_PP_LAUNCH(0, 4, 3, 1);
    }
    else if (dbg == 1) RL_PP_LAUNCH(1, 6, 3, 0);
    else if (dbg == 2) RL_PP_LAUNCH(2, 6, 3, 0);
    else if (dbg == 11) RL_PP_LAUNCH(11, 6, 3, 0);
    else if (dbg == 59) RL_PP_LAUNCH(59, 6, 3, 0);
    else if (dbg == 48) RL_PP_LAUNCH(48, 6, 3, 0);
    else if (dbg == 49) RL_PP_LAUNCH(49, 6, 3, 0);
    else RL_PP_LAUNCH(0, 6, 3, 0);
#undef RL_PP_LAUNCH
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
