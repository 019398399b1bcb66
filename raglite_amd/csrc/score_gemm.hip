// a6 for MANY queries at once (BASELINE cfg 5: B = 1000 against a 1.25 M-row shard): raw dots
//     S[b * ld + r] = Q[b] . E[r]
// as an exact-fp32 MFMA GEMM with the metric of src/raglite/_typing.py:123-134 applied in the epilogue (same
// formulas as scan.hip:transform_kernel); select.hip takes the top-k.  Batched generalisation of src/raglite/_search.py:69-72.
//
// Why a second kernel next to maxsim_stream.hip (mode 1): that one re-streams the corpus once per 32 queries, which is
// right up to ~64 queries (HBM-bound, AI = nq/2 flop/B) but at B = 1000 moves 164 GB for 2.56 TFLOP.  Here a
// 128-query x 128-row tile keeps AI at 32 flop/B against L2 and the corpus leaves HBM about once: the bound is the
// fp32 matrix pipe (157.3 TF peak; v_mfma_f32_16x16x4_f32 = 256 flop/clk/CU).
//
// Structure (same lessons as the stream kernel -- a wave that issues MFMAs must do nothing else):
//   * one persistent 384-thread workgroup per CU: waves 0-3 compute a 64 x 64 quadrant each (16 accumulator tiles = 64
//     VGPRs), waves 4-5 only issue LDS-DMA: wave 4 the E rows, wave 5 the Q rows of every K slab;
//   * K slabs of 32 (128 B per row): one slab = 128 E rows + 128 Q rows = 32 KiB, ring of 4 slabs in LDS, loaders 3
//     slabs ahead, ONE workgroup barrier per slab (slab g+3 overwrites the slot of slab g-1, which every compute wave
//     has provably consumed before it reaches the barrier of slab g);
//   * `global_load_lds_dwordx4` writes 64 lanes x 16 B contiguously, so rows sit unpadded at a 128-B pitch; bank
//     conflicts are avoided by swizzling on the GLOBAL side instead: 16-B chunk c of row r is stored at chunk position
//     c ^ ((r >> 1) & 7), and a fragment read (16 lanes = 16 rows, same chunk) then hits 16 different 16-B bank groups;
//   * per slab and wave: 16 ds_read_b128 (8 E + 8 Q fragments, each good for 4 k-steps) feed 128 MFMAs; the second half
//     of the reads lands while the first 64 MFMAs run;
//   * tiles are enumerated so that the 8 query tiles of one row tile run back to back on ONE XCD (blockIdx % 8 is the
//     XCD): E comes from HBM once and then from that XCD's L2;
//   * deterministic and position-independent: every (row, query) sums K in the same fixed order.
#include <cstdlib>

#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int GM = 128;                 // queries per tile
constexpr int GN = 128;                 // corpus rows per tile
constexpr int GK = 32;                  // K per slab
constexpr int NSLOT = 4;                // LDS ring
constexpr int OP_BYTES = 128 * GK * 4;  // one operand of one slab: 128 rows x 128 B = 16 KiB
constexpr int SLAB_BYTES = 2 * OP_BYTES;

// 8 fp32 values (two fragment reads) scaled by a power of two -> exact fp16 (hi, lo) pairs, as in maxsim_stream.hip:
// x*s = hi + lo, hi = fp16_rtz(x*s), lo = fp16_rtz(x*s - hi).
__device__ __forceinline__ void split8(const f32x4& A0, const f32x4& A1, float s, h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
        const float x0 = (u < 4 ? A0[u] : A1[u - 4]) * s, x1 = (u < 4 ? A0[u + 1] : A1[u - 3]) * s;
        const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);
        const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
        hi[u] = ph[0]; hi[u + 1] = ph[1];
        lo[u] = pl[0]; lo[u + 1] = pl[1];
    }
}

// SPLIT = the fp16 (hi, lo) arithmetic of include/raglite_hip.h (RL_ARITH_F16_SPLIT): per K slab and 16 x 16 tile three
// v_mfma_f32_16x16x32_f16 (16 cycles each) replace eight v_mfma_f32_16x16x4_f32 (32 cycles each); the corpus is scaled by
// the index' power of two `e_scale`, every query by one of its own (`q_scale[b]`, from query_scale_kernel), both undone on
// the accumulators before the metric is applied.
template <bool SPLIT>
__global__ __launch_bounds__(384) void score_gemm_kernel(const float* __restrict__ E, int64_t n_rows, int32_t dim,
                                                         const float* __restrict__ Q, int32_t B,
                                                         float* __restrict__ S, int64_t ld, int64_t n_tiles,
                                                         int32_t QT, const float* __restrict__ row_norm,
                                                         const float* __restrict__ row_sumsq,
                                                         const float* __restrict__ q_sumsq, int mode,
                                                         float e_scale, const float* __restrict__ q_scale) {
    __shared__ __attribute__((aligned(16))) char smem[NSLOT * SLAB_BYTES];
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();
    const int nslab = dim / GK;
    const int64_t G = gridDim.x, b = blockIdx.x;
    if (n_tiles <= b) return;
    const int my_tiles = (int)((n_tiles - b + G - 1) / G);
    const int total = my_tiles * nslab;  // K slabs this workgroup consumes, tile after tile
    // Tile it of this workgroup -> first corpus row, first query.  Linear id L = it * G + b; L % 8 is the XCD the
    // workgroup runs on; each XCD walks its own row tiles and takes all QT query tiles of one before the next.
    auto decode = [&](int it, int64_t& row0, int32_t& q0) {
        const int64_t L = (int64_t)it * G + b;
        const int64_t x = L & 7, j = L >> 3;
        q0 = (int32_t)(j % QT) * GM;
        row0 = ((j / QT) * 8 + x) * GN;
    };

    if (wv < 4) {
        // ================================ COMPUTE WAVE ==============================================================
        const int wy = wv >> 1, wx = wv & 1;  // quadrant: queries [64 wy, +64), rows [64 wx, +64)
        const int fj = lane & 15, kq = lane >> 4;
        // fragment (16 rows x 4 k-steps): lane (fj, kq) reads chunk kk * 4 + kq of row fj, stored at chunk ^ (fj >> 1)
        const uint32_t sw0 = (uint32_t)((kq ^ (fj >> 1)) * 16), sw1 = (uint32_t)(((4 + kq) ^ (fj >> 1)) * 16);
        const uint32_t e_off = (uint32_t)((64 * wx + fj) * 128), q_off = (uint32_t)(OP_BYTES + (64 * wy + fj) * 128);
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int it = 0, sl = 0;
        // Slab g's fragments: 16 ds_read_b128 (8 E + 8 Q; each E read is good for 4 fp32 k-steps / half an fp16 MFMA operand).
        auto read_slab = [&](int g, f32x4 (&ea)[2][4], f32x4 (&qa)[2][4]) {
            const char* base = smem + (g & (NSLOT - 1)) * SLAB_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint32_t sw = kk ? sw1 : sw0;
#pragma unroll
                for (int a = 0; a < 4; ++a) ea[kk][a] = *reinterpret_cast<const f32x4*>(base + e_off + a * 2048 + sw);
#pragma unroll
                for (int c = 0; c < 4; ++c) qa[kk][c] = *reinterpret_cast<const f32x4*>(base + q_off + c * 2048 + sw);
            }
        };
        auto tile_done = [&]() {
                // tile done: C/D layout of 16x16x4 -- lane (16 gq + j) holds rows 4 gq .. 4 gq + 3 of column (query) j
                int64_t row0;
                int32_t q0;
                decode(it, row0, q0);
                const int gq = lane >> 4, j = lane & 15;
                // Metric epilogue: the formulas (and operation order) of scan.hip:transform_kernel, so fused and
                // unfused paths give the same bits.
                f32x4 rn[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int64_t r = row0 + 64 * wx + 16 * a + 4 * gq;
                    const float* src = mode == SCAN_COSINE ? row_norm : row_sumsq;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        rn[a][u] = (mode == SCAN_COSINE || mode == SCAN_L2) ? src[r + u < n_rows ? r + u : n_rows - 1] : 0.f;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int32_t q = q0 + 64 * wy + 16 * c + j;
                    const float qss = (mode == SCAN_COSINE || mode == SCAN_L2) ? q_sumsq[q < B ? q : B - 1] : 0.f;
                    const float qn = sqrtf(qss);
                    [[maybe_unused]] float unscale = 1.f;
                    if constexpr (SPLIT) unscale = 1.0f / (e_scale * q_scale[q < B ? q : B - 1]);  // powers of two: exact
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int64_t r = row0 + 64 * wx + 16 * a + 4 * gq;
                        f32x4 v = acc[a][c];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float d = SPLIT ? v[u] * unscale : v[u];
                            if (mode == SCAN_COSINE) v[u] = 1.0f - (1.0f - d / (rn[a][u] * qn));
                            else if (mode == SCAN_DOT) v[u] = 1.0f + d;
                            else if (mode == SCAN_L2) v[u] = 1.0f - sqrtf(fmaxf(rn[a][u] + qss - 2.0f * d, 0.f));
                        }
                        if (q < B && r < n_rows) {
                            float* o = S + (int64_t)q * ld + r;
                            if (r + 3 < n_rows && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                                *reinterpret_cast<f32x4*>(o) = v;
                            } else {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (r + u < n_rows) o[u] = v[u];
                            }
                        }
                        acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
        };
        auto mma_slab = [&](const f32x4 (&ea)[2][4], const f32x4 (&qa)[2][4]) {
            if constexpr (SPLIT) {
                // lane (fj, kq) holds k = 4 kq .. 4 kq + 3 and 16 + 4 kq .. of its row: the same 8 positions in both operands.
                // The query operand arrives pre-split (query_presplit_kernel): chunk kq of a row's 128-B slab piece holds the 8 hi
                // halves of exactly those positions, chunk 4 + kq the 8 lo halves.
                h16x8 eh[4], el[4], qh[4], ql[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    __builtin_memcpy(&qh[c], &qa[0][c], 16);
                    __builtin_memcpy(&ql[c], &qa[1][c], 16);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) split8(ea[0][a], ea[1][a], e_scale, eh[a], el[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[a], qh[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el[a], qh[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[a], ql[c], acc[a][c], 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[kk][a][tt], qa[kk][c][tt], acc[a][c], 0, 0, 0);
            }
            if (++sl == nslab) {
                tile_done();
                sl = 0;
                ++it;
            }
        };
        if constexpr (SPLIT) {
            // Software pipeline over two register sets: slab g's fragments are read (16 ds_read_b128, ~0.6-0.9 k cycles of LDS
            // time) while slab g-1 is converted and multiplied -- with the fp16 MFMAs a slab's arithmetic is only ~1.2 k cycles,
            // so the reads no longer hide behind it as they do behind 4.1 k cycles of fp32 MFMAs.  Every read of slab g-1 has
            // completed (lgkmcnt(0)) before this wave reaches barrier g, which is what lets the loaders refill its slot.
            f32x4 eaA[2][4], qaA[2][4], eaB[2][4], qaB[2][4];
            int g = 0;
            for (; g + 1 < total; g += 2) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                read_slab(g, eaA, qaA);
                if (g > 0) mma_slab(eaB, qaB);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                read_slab(g + 1, eaB, qaB);
                mma_slab(eaA, qaA);
            }
            if (g < total) {  // odd number of slabs: one more read into A
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                read_slab(g, eaA, qaA);
                if (g > 0) mma_slab(eaB, qaB);
                mma_slab(eaA, qaA);
            } else if (total > 0) {
                mma_slab(eaB, qaB);
            }
        } else {
            for (int g = 0; g < total; ++g) {
                asm volatile("s_barrier" ::: "memory");  // slab g has landed (the loaders waited on their vmcnt)
                f32x4 ea[2][4], qa[2][4];
                read_slab(g, ea, qa);
                mma_slab(ea, qa);
            }
        }
        return;
    }

    // ==================================== LOADER WAVE (4: E rows, 5: Q rows) ========================================
    // 16 DMAs per slab: DMA i fills rows 8i .. 8i+7 (1 KiB of LDS); lane l -> row 8i + (l >> 3), chunk position l & 7,
    // which holds source chunk (l & 7) ^ ((row >> 1) & 7).  Rows past the matrix are clamped (their results are never
    // stored).  2 instructions per DMA: M0 and the load (per-lane offsets are per-tile constants).
    const int op = wv - 4;
    const char* const src = reinterpret_cast<const char*>(op == 0 ? E : Q);
    const int64_t lim = (op == 0 ? n_rows : (int64_t)B) - 1;
    const int64_t pitch = (int64_t)dim * 4;
    const uint32_t lds_op = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) +
                            (uint32_t)(op * OP_BYTES);
    uint32_t voff[16];
    const char* tile_base = src;
    int cur_it = -1;
    auto issue = [&](int gi) {
        int gg = gi < total ? gi : total - 1;  // past the end: harmless re-read that keeps the vmcnt bookkeeping uniform
        const int it = gg / nslab, sl = gg - it * nslab;
        if (it != cur_it) {  // wave-uniform: new tile -> per-lane row offsets relative to the tile's first row
            cur_it = it;
            int64_t row0;
            int32_t q0;
            decode(it, row0, q0);
            int64_t first = op == 0 ? row0 : (int64_t)q0;
            first = first < lim ? first : lim;
            tile_base = src + first * pitch;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 8 * i + (lane >> 3);
                int64_t rr = first + r;
                rr = rr < lim ? rr : lim;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                voff[i] = (uint32_t)((rr - first) * pitch) + (uint32_t)(c * 16);
            }
        }
        const char* base = tile_base + sl * (GK * 4);
        const uint32_t lds = lds_op + (uint32_t)((gi & (NSLOT - 1)) * SLAB_BYTES);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(lds), "n"(i * 1024),
                         "v"(voff[i]), "s"(base)
                         : "memory", "m0", "scc");
    };
    issue(0);
    issue(1);
    issue(2);
    for (int g = 0; g < total; ++g) {
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // in-order: at most slabs g+1, g+2 outstanding => slab g landed
        asm volatile("s_barrier" ::: "memory");
        issue(g + 3);  // into the slot of slab g-1: every compute wave consumed it before reaching this barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 rows x 256 queries per tile, fp16-split arithmetic only.  With the fp16 MFMAs the 128 x 128 kernel above is bound by
// the bytes a K slab brings into the CU (32 KiB per 2.0 k cycles), not by the matrix pipe: a
// 128 x 256 tile moves 48 KiB for twice the flops.  704 threads: waves 0-7 compute a 64 x 64 quadrant each
// (wx = wv & 1 rows, wy = wv >> 1 queries), wave 8 streams the E rows, waves 9 and 10 the pre-split query rows (128 each);
// ring of 3 slabs of 48 KiB, loaders 2 slabs ahead, one barrier per slab.  Three waves share a SIMD, so a compute wave
// must stay under 168 VGPRs (it needs ~150: 64 accumulators, 64 of fragments, one converted pair at a time).
constexpr int GM2 = 256;
constexpr int NSLOT2 = 3;
constexpr int OPQ2_BYTES = GM2 * GK * 4;                 // 32 KiB
constexpr int SLAB2_BYTES = OP_BYTES + OPQ2_BYTES;       // 48 KiB

__global__ __launch_bounds__(704) void score_gemm256_kernel(const float* __restrict__ E, int64_t n_rows, int32_t dim,
                                                            const float* __restrict__ Qs, int32_t B,
                                                            float* __restrict__ S, int64_t ld, int64_t n_tiles,
                                                            int32_t QT, const float* __restrict__ row_norm,
                                                            const float* __restrict__ row_sumsq,
                                                            const float* __restrict__ q_sumsq, int mode,
                                                            float e_scale, const float* __restrict__ q_scale) {
    __shared__ __attribute__((aligned(16))) char smem[NSLOT2 * SLAB2_BYTES];
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();
    const int nslab = dim / GK;
    const int64_t G = gridDim.x, b = blockIdx.x;
    if (n_tiles <= b) return;
    const int my_tiles = (int)((n_tiles - b + G - 1) / G);
    const int total = my_tiles * nslab;
    auto decode = [&](int it, int64_t& row0, int32_t& q0) {
        const int64_t L = (int64_t)it * G + b;
        const int64_t x = L & 7, j = L >> 3;
        q0 = (int32_t)(j % QT) * GM2;
        row0 = ((j / QT) * 8 + x) * GN;
    };

    if (wv < 8) {
        // ================================ COMPUTE WAVE ==============================================================
        const int wx = wv & 1, wy = wv >> 1;  // rows [64 wx, +64), queries [64 wy, +64)
        const int fj = lane & 15, kq = lane >> 4;
        const uint32_t sw0 = (uint32_t)((kq ^ (fj >> 1)) * 16), sw1 = (uint32_t)(((4 + kq) ^ (fj >> 1)) * 16);
        const uint32_t e_off = (uint32_t)((64 * wx + fj) * 128), q_off = (uint32_t)(OP_BYTES + (64 * wy + fj) * 128);
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int it = 0, sl = 0;
        for (int g = 0; g < total; ++g) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // slab g has landed
            const char* base = smem + (g % NSLOT2) * SLAB2_BYTES;
            h16x8 qh[4], ql[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                qh[c] = *reinterpret_cast<const h16x8*>(base + q_off + c * 2048 + sw0);
                ql[c] = *reinterpret_cast<const h16x8*>(base + q_off + c * 2048 + sw1);
            }
            f32x4 e0[4], e1[4];  // all 16 reads of the slab are in flight before the first conversion
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                e0[a] = *reinterpret_cast<const f32x4*>(base + e_off + a * 2048 + sw0);
                e1[a] = *reinterpret_cast<const f32x4*>(base + e_off + a * 2048 + sw1);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                h16x8 eh, el;
                split8(e0[a], e1[a], e_scale, eh, el);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, qh[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, qh[c], acc[a][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, ql[c], acc[a][c], 0, 0, 0);
            }
            if (++sl == nslab) {
                int64_t row0;
                int32_t q0;
                decode(it, row0, q0);
                const int gq = lane >> 4, j = lane & 15;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int32_t q = q0 + 64 * wy + 16 * c + j;
                    const float qss = (mode == SCAN_COSINE || mode == SCAN_L2) ? q_sumsq[q < B ? q : B - 1] : 0.f;
                    const float qn = sqrtf(qss);
                    const float unscale = 1.0f / (e_scale * q_scale[q < B ? q : B - 1]);  // powers of two: exact
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int64_t r = row0 + 64 * wx + 16 * a + 4 * gq;
                        const float* src = mode == SCAN_COSINE ? row_norm : row_sumsq;
                        f32x4 v = acc[a][c];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float rn = (mode == SCAN_COSINE || mode == SCAN_L2) ? src[r + u < n_rows ? r + u : n_rows - 1] : 0.f;
                            const float d = v[u] * unscale;
                            if (mode == SCAN_COSINE) v[u] = 1.0f - (1.0f - d / (rn * qn));
                            else if (mode == SCAN_DOT) v[u] = 1.0f + d;
                            else if (mode == SCAN_L2) v[u] = 1.0f - sqrtf(fmaxf(rn + qss - 2.0f * d, 0.f));
                        }
                        if (q < B && r < n_rows) {
                            float* o = S + (int64_t)q * ld + r;
                            if (r + 3 < n_rows && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                                *reinterpret_cast<f32x4*>(o) = v;
                            } else {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                    if (r + u < n_rows) o[u] = v[u];
                            }
                        }
                        acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                }
                sl = 0;
                ++it;
            }
        }
        return;
    }

    // ==================================== LOADER WAVES (8: E rows; 9, 10: query rows 0-127, 128-255) ================
    const int op = wv == 8 ? 0 : 1, half = wv == 10 ? 1 : 0;
    const char* const src = reinterpret_cast<const char*>(op == 0 ? E : Qs);
    const int64_t lim = (op == 0 ? n_rows : (int64_t)B) - 1;
    const int64_t pitch = (int64_t)dim * 4;
    const uint32_t lds_op = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) +
                            (uint32_t)(op * OP_BYTES + half * 16 * 1024);
    uint32_t voff[16];
    const char* tile_base = src;
    int cur_it = -1;
    auto issue = [&](int gi) {
        int gg = gi < total ? gi : total - 1;
        const int it = gg / nslab, sl = gg - it * nslab;
        if (it != cur_it) {
            cur_it = it;
            int64_t row0;
            int32_t q0;
            decode(it, row0, q0);
            int64_t first = op == 0 ? row0 : (int64_t)q0;
            first = first < lim ? first : lim;
            tile_base = src + first * pitch;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 128 * half + 8 * i + (lane >> 3);
                int64_t rr = first + r;
                rr = rr < lim ? rr : lim;
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                voff[i] = (uint32_t)((rr - first) * pitch) + (uint32_t)(c * 16);
            }
        }
        const char* base = tile_base + sl * (GK * 4);
        const uint32_t lds = lds_op + (uint32_t)((gi % NSLOT2) * SLAB2_BYTES);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(lds), "n"(i * 1024),
                         "v"(voff[i]), "s"(base)
                         : "memory", "m0", "scc");
    };
    issue(0);
    issue(1);
    for (int g = 0; g < total; ++g) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // in-order: at most slab g+1 outstanding => slab g landed
        asm volatile("s_barrier" ::: "memory");
        issue(g + 2);  // into the slot of slab g-1: every compute wave consumed it before reaching this barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Sum of squares of every query, in transform_kernel's summation order (256 strided partial sums, wave butterflies,
// ((p0+p1)+(p2+p3))), so the fused epilogue reproduces the unfused path bit for bit.
__global__ __launch_bounds__(256) void query_sumsq_kernel(const float* __restrict__ queries, int dim,
                                                           float* __restrict__ q_sumsq) {
    __shared__ float part[4];
    const int b = blockIdx.x;
    float ss = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        const float v = queries[(int64_t)b * dim + c];
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) q_sumsq[b] = (part[0] + part[1]) + (part[2] + part[3]);
}
// SPLIT arithmetic, query side, once per batch: q_scale[b] = the power of two that brings query b's largest |element|
// into [2^13, 2^14), and Qs = the scaled query as fp16 (hi, lo) pairs in the GEMM's slab layout -- the 128 B that hold
// k = 32 s .. 32 s + 31 of a row become 8 chunks of 16 B: chunk kq (0..3) = hi of k = 32 s + {4 kq .. 4 kq + 3, 16 + 4 kq ..
// 16 + 4 kq + 3}, chunk 4 + kq = the lo halves of the same positions: exactly what lane (., kq) of a compute wave holds of E.
// Same bytes per row as the fp32 query, so the loaders move it unchanged.
__global__ __launch_bounds__(256) void query_presplit_kernel(const float* __restrict__ queries, int dim, float* __restrict__ q_scale,
                                                             uint4* __restrict__ Qs) {
    __shared__ float part[4];
    const int b = blockIdx.x;
    const float* q = queries + (int64_t)b * dim;
    float mx = 0.f;
    for (int c = threadIdx.x; c < dim; c += 256) mx = fmaxf(mx, fabsf(q[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);
    const float sc = ldexpf(1.f, 14 - (ex > -100 ? ex : -100));
    if (threadIdx.x == 0) q_scale[b] = sc;
    // one thread per (slab, kq): 8 elements -> one hi chunk and one lo chunk
    for (int t = threadIdx.x; t < dim / 8; t += 256) {
        const int sl = t >> 2, kq = t & 3;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(q + 32 * sl + 4 * kq), v1 = *reinterpret_cast<const f32x4*>(q + 32 * sl + 16 + 4 * kq);
        h16x8 hi, lo;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = (u < 4 ? v0[u] : v1[u - 4]) * sc;
            const _Float16 h = (_Float16)x;
            hi[u] = h;
            lo[u] = (_Float16)(x - (float)h);
        }
        uint4 a, c;
        __builtin_memcpy(&a, &hi, 16);
        __builtin_memcpy(&c, &lo, 16);
        uint4* row = Qs + (int64_t)b * (dim / 4) + sl * 8;
        row[kq] = a;
        row[4 + kq] = c;
    }
}
}  // namespace

// |q|^2 per query in transform_kernel's summation order (shared with maxsim_gemm.hip's row-score mode)
int launch_query_sumsq(const float* Q, int32_t nb, int32_t dim, float* out, hipStream_t s) {
    hipLaunchKernelGGL(query_sumsq_kernel, dim3(nb), dim3(256), 0, s, Q, (int)dim, out);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Similarity (metric `mode`, scan.hip conventions) of nb queries against every row; dim % 32 == 0, 16-B aligned
// operands.  q_sumsq_scratch: device float[nb] (cosine / l2 only).
// split_scale > 0: fp16-split arithmetic with the corpus scaled by that power of two; q_sumsq_scratch then holds
// score_gemm_scratch_floats(nb, dim) floats: q_sumsq, q_scale and the pre-split queries.
size_t score_gemm_scratch_floats(int32_t nb, int32_t dim, bool split) { return split ? (size_t)nb * (dim + 8) + 8 : (size_t)nb; }
int launch_score_gemm(const float* E, int64_t n_rows, int32_t dim, const float* Q, int32_t nb, float* scores,
                      int64_t ld, const float* row_norm, const float* row_sumsq, float* q_sumsq_scratch, int mode,
                      int n_cu, hipStream_t s, float split_scale) {
    if (nb < 1 || n_rows < 1 || dim < GK || dim % GK != 0) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(E) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    if ((mode == SCAN_COSINE && !row_norm) || (mode == SCAN_L2 && !row_sumsq)) return RL_ERR_INVALID;
    const int64_t RT = (n_rows + GN - 1) / GN;
    const int32_t QT = (nb + GM - 1) / GM;
    const int64_t n_tiles = ((RT + 7) / 8) * 8 * QT;  // row tiles padded to a multiple of 8 (one residue class per XCD)
    const int grid = (int)std::min<int64_t>(n_cu > 0 ? n_cu : 256, n_tiles);
    if (!q_sumsq_scratch && (mode == SCAN_COSINE || mode == SCAN_L2 || split_scale > 0.f)) return RL_ERR_INVALID;
    if (mode == SCAN_COSINE || mode == SCAN_L2)
        hipLaunchKernelGGL(query_sumsq_kernel, dim3(nb), dim3(256), 0, s, Q, (int)dim, q_sumsq_scratch);
    if (split_scale > 0.f) {
        float* q_scale = q_sumsq_scratch + nb;
        float* Qs = q_sumsq_scratch + (((size_t)2 * nb + 7) & ~(size_t)7);  // 32-B aligned behind q_sumsq and q_scale
        if (reinterpret_cast<uintptr_t>(Qs) & 15) return RL_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(query_presplit_kernel, dim3(nb), dim3(256), 0, s, Q, (int)dim, q_scale, reinterpret_cast<uint4*>(Qs));
        static const bool tile128 = exp_env("RAGLITE_GEMM_TILE128") != nullptr;  // A/B switch
        if (!tile128 && nb > GM) {  // 128 x 256 tiles: 25 % fewer bytes into the CU per flop
            const int32_t QT2 = (nb + GM2 - 1) / GM2;
            const int64_t n_tiles2 = ((RT + 7) / 8) * 8 * QT2;
            const int grid2 = (int)std::min<int64_t>(n_cu > 0 ? n_cu : 256, n_tiles2);
            hipLaunchKernelGGL(score_gemm256_kernel, dim3(grid2), dim3(704), 0, s, E, n_rows, dim, Qs, nb, scores, ld, n_tiles2, QT2,
                               row_norm, row_sumsq, q_sumsq_scratch, mode, split_scale, q_scale);
            RL_HIP(hipGetLastError());
            return RL_OK;
        }
        hipLaunchKernelGGL(score_gemm_kernel<true>, dim3(grid), dim3(384), 0, s, E, n_rows, dim, Qs, nb, scores, ld, n_tiles, QT,
                           row_norm, row_sumsq, q_sumsq_scratch, mode, split_scale, q_scale);
    } else {
        hipLaunchKernelGGL(score_gemm_kernel<false>, dim3(grid), dim3(384), 0, s, E, n_rows, dim, Q, nb, scores, ld, n_tiles, QT,
                           row_norm, row_sumsq, q_sumsq_scratch, mode, 0.f, nullptr);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
