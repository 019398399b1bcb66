// a9 (metric shape) and batched a6: the MFMA streaming tile kernel, dim in {128, 256, 384, 512, 768, 1024}
// (the numbers below are for dim = 1024, bge-m3, the reference's default embedder).
//
//   mode 0 (MaxSim):  out[c]          = sum_{i<nq} max_{j in chunk c} Q[i].D[j]      (nq <= 32)
//   mode 1 (rows):    out[i*ld + row] = Q[i].D[row]   (raw dots; scan.hip:transform_kernel applies the metric of
//                                       src/raglite/_typing.py:123-134 afterwards)
// Multi-query generalisation of src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174.
//
// Roofline: one corpus pass moves 4*N*1024 B from HBM and needs 2*nq*N*1024 flop of exact
// fp32 MFMA (157.3 TF peak).  Per 16-row tile and CU that is 64 KiB = ~6.4 k cycles at 6.3 TB/s / 256 CUs against
// 128 x 32 = 4.1 k cycles of v_mfma_f32_16x16x4_f32 per SIMD at nq = 32: HBM-bound, but only if the matrix pipe
// and the memory pipeline really run side by side.
//
// Design (what the measurements forced -- see DESIGN.md section 4.1 and profiles/r01_*):
//  * one 512-thread workgroup per CU (LDS-limited), persistent over a contiguous, chunk-aligned row range, so
//    per-chunk maxima never cross workgroups and there is no inter-workgroup hand-off;
//  * WAVE SPECIALISATION, shaped by one hardware fact: while a wave issues back-to-back v_mfma_f32_16x16x4_f32, every
//    other wave on that SIMD gets an issue slot only every ~30-60 cycles (s_setprio does not help; s_nop between the
//    MFMAs only slows them).  All four SIMDs carry a compute wave, so everything else must be FEW instructions, or
//    must run while the compute waves are not multiplying:
//      waves 0-3  COMPUTE, one per SIMD: K-split (wave w owns columns [256w, 256w+256); its 32 x 256 slice of Q
//                 lives in 128 VGPRs as the MFMA B operand for the whole kernel, so Q costs no LDS traffic).
//                 Per tile: 16 ds_read_b128 (A fragments), 128 MFMAs, one 2-KiB K-partial write.  Nothing else.
//      waves 4-5  LOAD: stream D HBM -> LDS with `global_load_lds_dwordx4 ... nt` (no VGPR round trip; one instruction
//                 = 1 KiB of a row) into two row-major 16-row stages, 1.75 tiles ahead.  Hand-issued: 1.5 instructions
//                 per DMA (the compiler's 6 per DMA made the loaders the bottleneck at ~5.8 k cycles per tile).  Rows
//                 sit at a (row + 32 B) pitch: ds_read_b128 serves 8 rows x 2 k-groups per cycle, which this pitch
//                 spreads over all 64 banks (measured 236 B/clk/CU; +16 B gives 128, scripts/micro/lds_b128.hip).
//                 Loader 0 also brings the tile's chunk ordinals into a small LDS ring.
//                 In the B1..B2 window (below) the loaders do the K-reduction of the previous tile's partials.
//      wave 6     EPILOGUE: segmented running max over the tile's rows (one v_max per row, wave-uniform branch at
//                 chunk ends), DPP-butterfly sum over the query columns per finished chunk, one store per tile.  The
//                 walk runs in the B1..B2 window; in the MFMA phase the wave only parks.  Wave 7 is idle (barriers).
//  * two workgroup barriers per tile hand the stages back and forth:
//      B1(t): tile t has landed (loaders waited on their own vmcnt) and the K-partials of tile t-1 are in LDS;
//      B2(t): every compute wave holds tile t in registers -> stage t&1 may be refilled with tile t+2; the reduced
//             tile t-1 is in ST.
//    The B1..B2 window is the only time the compute waves are not issuing MFMAs (they read A fragments, ~0.5 k
//    cycles of LDS bandwidth), so it hosts the K-reduction (loaders) and the row walk of tile t-2 (wave 6).
//    The barriers retire LDS operations only (`s_waitcnt lgkmcnt(0); s_barrier`), never VMEM: the DMAs stay in
//    flight across them, counted with `s_waitcnt vmcnt(N)`.
//  * deterministic: fixed K order inside a wave (fp32 MFMA is bitwise an ordered fmaf chain), fixed
//    ((p0+p1)+(p2+p3)) across waves, fixed butterfly order of the per-chunk sum.
// Measured on MI355X (32 x 1M x 1024, ragged chunks): 0.66 ms per corpus pass = 6.2 TB/s = 78 % of the 8 TB/s peak --
// the rate of the MFMA-free single-query scan kernel on the same box (scripts/kernel_ab.py); history 0.98 -> 0.81 ->
// 0.66 ms and tile timelines in DESIGN.md section 4.1 / profiles/r01_tile_timeline*.txt.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int TR = 16;      // rows per tile (one 16x16x4 MFMA row block)
constexpr int NSTAGE = 2;
// Compile-time geometry for dim = 4 * KW (KW = columns per compute wave, a multiple of 16).
template <int KW, bool F16 = false>
struct Geo {
    static constexpr int DIM = 4 * KW;
    static constexpr int ELT = F16 ? 2 : 4;                // corpus element: fp32, or fp16 storage
    static constexpr int QBYTES = KW * ELT;                // one K quarter of a row (what one compute wave consumes)
    static constexpr int ROWB = DIM * ELT;                 // one corpus row
    static constexpr int PITCH = ROWB + 32;                // +32 B, MEASURED (scripts/micro/lds_b128.hip): ds_read_b128 serves 8 rows x
                                                           // 2 k-groups per cycle, so the row stride must be 8 banks (mod 64):
                                                           // +32 B -> 236 B/clk/CU, +16 B -> 128 (2-way conflicts), +0 -> 32
    static constexpr int STAGE = TR * PITCH;               // one tile, row-major: [16 rows][DIM fp32 + pad]
    static constexpr int NCH = (ROWB + 1023) / 1024;       // 1-KiB DMA instructions per row (the last may be half)
    static constexpr int KSTEPS = F16 ? KW / 32 : KW / 16; // ds_read_b128 per lane per tile: 4 fp32 k-steps of 4, or one
                                                           // 16x16x32 f16 MFMA, each
    static constexpr int OFF_RED = NSTAGE * STAGE;
    static constexpr int RED_BYTES = 4 * 2 * 64 * 16;      // 8192: K-partial exchange buffer (4 waves x 2 query halves)
    static constexpr int OFF_ST = OFF_RED + RED_BYTES;     // K-reduced tile, transposed: [32 query columns][20] fp32
    static constexpr int ST_PITCH = 20;                    // 16 rows + 4 pad floats: 80-B columns, b128-aligned
    static constexpr int OFF_ORD = OFF_ST + 32 * ST_PITCH * 4;    // chunk ordinals of the tiles' rows: ring of 4 x 64 int32
    static constexpr int LDS_TOTAL = OFF_ORD + 4 * 256;           // 143360 B at KW = 256
};

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// Workgroup barrier that also retires this wave's LDS / SMEM operations, but never its VMEM (the DMAs stay in flight).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

template <int KW, int NQT, int MODE, bool TRACE = false, int SPLR = 6, bool F16 = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void maxsim_stream_kernel(const float* __restrict__ D, int64_t n_rows,
                                                                 const float* __restrict__ Q, int nq,
                                                                 const int32_t* __restrict__ row_to_chunk,
                                                                 const int64_t* __restrict__ chunk_offsets,
                                                                 int64_t n_chunks, float* __restrict__ out,
                                                                 int64_t ld, unsigned long long* trace, float e_scale,
                                                                 int64_t q_batch_stride, int64_t out_batch_stride, StreamSecondJob j2) {
    static_assert(!(F16 && SPLIT), "SPLIT is a way to multiply an fp32-stored corpus");
    // j2.D != nullptr: TWO jobs in one launch -- grid row 1 runs the same queries over another matrix into another output, behind its own
    // run-if flag (the guarded full-precision pass of the half-bytes row search rides on the launch that re-scores its candidates: a guarded
    // launch that returns at once still costs 4.6 us of a 0.37 ms single-query search)
    if (j2.D && blockIdx.y == 1) {
        D = j2.D;
        n_rows = j2.n_rows;
        out = j2.out;
        ld = j2.ld;
        trace = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(j2.run_if));
    }
    // gridDim.y > 1: ONE launch for a batch of queries (the guarded full-precision fallback of a MaxSim batch over an index that keeps no
    // pre-split image): grid row y scores query y -- its own vectors, its own output row
    Q += (int64_t)blockIdx.y * q_batch_stride;
    out += (int64_t)blockIdx.y * out_batch_stride;
    if constexpr (!TRACE) {  // production build: `trace` carries an optional run-if flag (a guarded fallback launch returns at once)
        if (trace && __builtin_amdgcn_readfirstlane((int)*reinterpret_cast<const uint32_t*>(trace)) == 0) return;  // whole grid
    }
    using G_ = Geo<KW, F16>;
    constexpr int SD = G_::DIM, PITCH = G_::PITCH, STAGE = G_::STAGE, KSTEPS = G_::KSTEPS, NCH = G_::NCH, ROWB = G_::ROWB;
    constexpr int OFF_RED = G_::OFF_RED, OFF_ST = G_::OFF_ST, ST_PITCH = G_::ST_PITCH, OFF_ORD = G_::OFF_ORD;
    __shared__ __attribute__((aligned(16))) char smem[G_::LDS_TOTAL + (TRACE ? 4096 : 0)];
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();          // 0..7
    // Optional tile timeline (diagnostic build only, RAGLITE_HIP_TRACE=1): workgroup 7, tiles 100..107, 8
    // s_memtime stamps per wave per tile, kept in LDS (a global store per stamp would queue behind the loaders' DMAs
    // and distort the timeline) and copied out by each wave before it exits.  Compiled out of the production kernel.
    int nt_trace = 0;
    auto stamp = [&](int t, int k) {
        if constexpr (TRACE) {
            // rows of the dump: tiles 100..105, then the workgroup's first and last tile (whole-kernel span)
            const int slot = (t >= 100 && t < 106) ? t - 100 : (t == 0 ? 6 : (t == nt_trace - 1 ? 7 : -1));
            if (blockIdx.x == 7 && slot >= 0 && lane == 0)
                reinterpret_cast<unsigned long long*>(smem + G_::LDS_TOTAL)[(slot * 8 + wv) * 8 + k] =
                    __builtin_amdgcn_s_memtime();
        }
    };
    auto dump_trace = [&]() {
        if constexpr (TRACE) {
            if (blockIdx.x == 7) {
                const int i = ((lane >> 3) * 8 + wv) * 8 + (lane & 7);
                trace[i] = reinterpret_cast<unsigned long long*>(smem + G_::LDS_TOTAL)[i];
            }
        }
    };
    if constexpr (TRACE) {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) reinterpret_cast<unsigned long long*>(smem + G_::LDS_TOTAL)[i] = 0;
        __syncthreads();
    }
    const int w = wv & 3;              // K quarter this wave computes (waves 0-3) or feeds (waves 4-7)
    const bool is_loader = wv >= 4;    // wave-uniform
    const int64_t G = gridDim.x, b = blockIdx.x;

    int64_t r_lo, r_hi;
    if constexpr (MODE == 0) {
        // First chunk boundary at or after row t, through the row -> chunk map: two dependent loads instead of a
        // 17-step binary search over chunk_offsets (which cost ~5 us of every launch -- 6 % of a 125 k-row shard's pass).
        auto boundary = [&](int64_t t) -> int64_t {
            if (t <= 0) return 0;
            if (t >= n_rows) return n_rows;
            const int32_t c = row_to_chunk[t];
            const int64_t c0 = chunk_offsets[c], c1 = chunk_offsets[c + 1];
            return c0 == t ? t : c1;
        };
        r_lo = boundary((n_rows * b) / G);
        r_hi = (b + 1 == G) ? n_rows : boundary((n_rows * (b + 1)) / G);
    } else {
        const int64_t tiles = (n_rows + TR - 1) / TR;
        r_lo = ((tiles * b) / G) * TR;
        r_hi = ((tiles * (b + 1)) / G) * TR;
        if (r_hi > n_rows) r_hi = n_rows;
    }
    r_lo = uniform_i64(r_lo);
    r_hi = uniform_i64(r_hi);
    const int nt = (int)((r_hi - r_lo + TR - 1) / TR);
    nt_trace = nt;
    if (nt <= 0) return;  // whole workgroup: no barrier is skipped by a subset of its waves
    const int32_t last_row = (int32_t)(n_rows - 1);  // n_rows < 2^31 (checked by rl_index_create)
    const int32_t r_lo32 = (int32_t)r_lo;
    char* const red = smem + OFF_RED;
    float* const ST = reinterpret_cast<float*>(smem + OFF_ST);
    constexpr int NQC = 16 * NQT;

    if (!is_loader) {
        // ================================ COMPUTE WAVE ==========================================================
        // Q slice as MFMA B fragments: lane (j = lane & 15, kq = lane >> 4) holds, for MFMA 4*mm + tt,
        // Q[16*h + j][256*w + 16*mm + 4*kq + tt]  (a fixed permutation of K inside the wave).
        const int fj = lane & 15, kq = lane >> 4;
        // fp32 corpus: qreg[h][4 mm + tt] feeds v_mfma_f32_16x16x4_f32.  fp16 corpus: the query slice is split into
        // fp16 hi + lo halves (q = hi + lo to ~2^-22 relative; lo == 0 for the reference's fp16-valued queries) that feed
        // v_mfma_f32_16x16x32_f16 -- fp16 x fp16 products are exact in fp32, so nothing is lost against the stored data.
        constexpr bool H = F16 || SPLIT;      // the MFMAs are v_mfma_f32_16x16x32_f16
        constexpr int MS = KW / 32;           // ... MS of them per query tile and operand pair
        [[maybe_unused]] float qreg[H ? 1 : NQT][H ? 1 : KW / 4];
        [[maybe_unused]] h16x8 qhi[H ? NQT : 1][H ? MS : 1], qlo[H ? NQT : 1][H ? MS : 1];
        [[maybe_unused]] bool any_lo = false;
        [[maybe_unused]] float q_unscale = 1.f;
        if constexpr (SPLIT || F16) {
            // SPLIT: x = hi + lo with hi, lo fp16 (22 significant bits; fp16 x fp16 products are exact in the MFMA's fp32
            // accumulators), so  e.q ~= eh.qh + (el.qh + eh.ql)  costs 3 fp16 MFMAs of 16 cycles where the exact path
            // issues 8 fp32 MFMAs of 32.  Both operands are first brought to the top of fp16's range by powers of two
            // (exact, undone on the K-partials) so that lo keeps its 11 bits instead of going subnormal: the slice of Q by
            // a scale of its own, the corpus by `e_scale` (chosen from the largest row norm when the index is built).
            float mx = 0.f;
#pragma unroll
            for (int h = 0; h < NQT; ++h) {
                const int qi = 16 * h + fj;
                if (qi < nq)
                    for (int c = 0; c < KW / 4; c += 4) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(Q + (int64_t)qi * SD + KW * w + KW / 4 * kq + c);
                        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            int ex = 0;
            if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
            const float q_scale = ldexpf(1.f, 14 - ex);           // |q| * q_scale < 2^14
            q_unscale = F16 ? ldexpf(1.f, ex - 14) : ldexpf(1.f, ex - 14) / e_scale;  // stored halves are not rescaled
#pragma unroll
            for (int h = 0; h < NQT; ++h) {
                const int qi = 16 * h + fj;
                const int qc_ = qi < nq ? qi : nq - 1;
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    // fp32 corpus: k = 16 (2m + (u >> 2)) + 4 kq + (u & 3) (two fp32 fragment reads); fp16-stored corpus:
                    // k = 32 m + 8 kq + u (one read of 8 stored halves)
                    const float* qp = Q + (int64_t)qc_ * SD + KW * w + 32 * m + (F16 ? 8 : 4) * kq;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp), v1 = *reinterpret_cast<const f32x4*>(qp + (F16 ? 4 : 16));
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float x = qi < nq ? (u < 4 ? v0[u] : v1[u - 4]) * q_scale : 0.f;
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
        for (int h = 0; h < ((SPLIT || F16) ? 0 : NQT); ++h) {
            const int qi = 16 * h + fj;
            const int qc_ = qi < nq ? qi : nq - 1;  // clamped load, zeroed below: padded query vectors add 0
#pragma unroll
            for (int mm = 0; mm < KSTEPS; ++mm) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Q + (int64_t)qc_ * SD + KW * w + 16 * mm + 4 * kq);
                if (qi >= nq) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                qreg[h][4 * mm + 0] = v[0]; qreg[h][4 * mm + 1] = v[1];
                qreg[h][4 * mm + 2] = v[2]; qreg[h][4 * mm + 3] = v[3];
            }
        }
        const char* const a_base = smem + fj * PITCH + w * G_::QBYTES + kq * 16;
        for (int t = 0; t < nt; ++t) {
            stamp(t, 0);
            wg_barrier();  // B1(t): tile t is in stage t&1
            stamp(t, 1);
            f32x4 a[KSTEPS];  // 16 B per lane and step: 4 fp32 (k = 16 mm + 4 kq ..) or 8 fp16 (k = 32 mm + 8 kq ..)
            const char* ap = a_base + (t & 1) * STAGE;
#pragma unroll
            for (int mm = 0; mm < KSTEPS; ++mm) a[mm] = *reinterpret_cast<const f32x4*>(ap + mm * 64);
            stamp(t, 2);
            wg_barrier();  // B2(t): (after lgkmcnt(0)) the stage may be refilled
            stamp(t, 3);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[NQT];
#pragma unroll
            for (int h = 0; h < NQT; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                f32x4 acl[NQT];  // the two cross terms (2^-11 of the main one), summed apart
#pragma unroll
                for (int h = 0; h < NQT; ++h) acl[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    h16x8 eh, el;
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        const f32x2 x = (f32x2){a[2 * m + (u >> 2)][u & 3], a[2 * m + (u >> 2)][(u & 3) + 1]} * e_scale;
                        const auto ph = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);  // truncation: the residual is exact in fp32
                        const auto pl = __builtin_amdgcn_cvt_pkrtz(x[0] - (float)ph[0], x[1] - (float)ph[1]);
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
#pragma unroll
                for (int mm = 0; mm < KSTEPS; ++mm)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int h = 0; h < NQT; ++h)
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mm][tt], qreg[h][4 * mm + tt], acc[h], 0, 0, 0);
            } else {
                f32x4 acl[NQT];  // the lo term, summed apart (as in the SPLIT path and in maxsim_stream2_kernel: same bits)
#pragma unroll
                for (int h = 0; h < NQT; ++h) acl[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mm = 0; mm < KSTEPS; ++mm) {
                    h16x8 af;
                    __builtin_memcpy(&af, &a[mm], 16);
#pragma unroll
                    for (int h = 0; h < NQT; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qhi[h][mm], acc[h], 0, 0, 0);
                    if (any_lo) {
#pragma unroll
                        for (int h = 0; h < NQT; ++h)
                            acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qlo[h][mm], acl[h], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int h = 0; h < NQT; ++h) acc[h] = (acc[h] + acl[h]) * q_unscale;
            }
            stamp(t, 4);
#pragma unroll
            for (int h = 0; h < NQT; ++h) *reinterpret_cast<f32x4*>(red + ((w * 2 + h) * 64 + lane) * 16) = acc[h];
        }
        wg_barrier();  // B1(nt): publishes the K-partials of the last tile
        wg_barrier();  // B2(nt): (the loaders reduce them in between)
        dump_trace();
        return;
    }

    if (wv < 6) {
        // ==================================== LOADER WAVE (4, 5) ==================================================
        // Two of the four extra waves stream and never do anything else: loader l feeds rows 8l..8l+7 of every tile,
        // NCH 1-KiB LDS-DMA instructions per row.  What bounds them is INSTRUCTION ISSUE, not memory: a wave that shares a
        // SIMD with a compute wave gets about one issue slot per 32-cycle MFMA while that wave is in its MFMA phase
        // (measured: the compiler's 6 instructions per DMA -- 64-bit address arithmetic, row clamp, M0 -- took ~5.8 k
        // cycles per tile even with 7/8 of the chip idle).  So the steady-state path is hand-issued and costs
        // 1.5 instructions per DMA: per row one SALU write of M0 (+ the hazard nop) and NCH DMAs that share one
        // precomputed 32-bit lane offset, a wave-uniform 64-bit tile base, and the instruction's immediate offset,
        // which the hardware adds to BOTH the global and the LDS address (hence the row-major stage layout).
        const int lq = wv - 4;
        const char* const src = reinterpret_cast<const char*>(D);
        const uint32_t lane_off = 16u * lane;
        uint32_t voff[8];  // byte offset of (row 8l + i, lane) inside a tile
#pragma unroll
        for (int i = 0; i < 8; ++i) voff[i] = (uint32_t)((8 * lq + i) * ROWB) + lane_off;
        const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) +
                              (uint32_t)(8 * lq * PITCH);
        constexpr int TAIL_LANES = (ROWB % 1024) / 16;  // lanes of a row's last DMA when the row is not a multiple of 1 KiB
        constexpr bool HALF_TAIL = TAIL_LANES != 0;
        // Rows i0..i1-1 (of this loader's 8) of tile t -> stage t&1.
        const uint32_t lds_ord = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem) + OFF_ORD;
        const uint32_t voff_ord = 4u * lane;
        constexpr bool ORD = MODE == 0;  // MaxSim: loader 0 also brings the tile's chunk ordinals (64 int32 from its first row)
        auto dma_rows = [&](int t, auto I0_, auto I1_) {
            constexpr int i0 = decltype(I0_)::value, i1 = decltype(I1_)::value;
            const int32_t row0 = r_lo32 + t * TR;
            if constexpr (ORD && i0 == 0) {
                if (lq == 0) {  // wave-uniform
                    const int32_t rowc = row0 < last_row ? row0 : last_row;
                    const int32_t* rc = row_to_chunk + rowc;
                    const uint32_t dst = lds_ord + (uint32_t)((t & 3) * 256);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(voff_ord), "s"(rc)
                                 : "memory", "m0");
                }
            }
            if (row0 + TR - 1 <= last_row) {  // wave-uniform: the whole tile is inside the corpus
                const char* base = src + (int64_t)row0 * ROWB;
                const uint32_t lds = lds0 + (uint32_t)((t & 1) * STAGE);
#pragma unroll
                for (int i = i0; i < i1; ++i) {
                    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0" ::"s"(lds), "n"(i * PITCH) : "memory", "m0", "scc");
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (HALF_TAIL && c == NCH - 1) {
                            if (lane < TAIL_LANES)
                                asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(voff[i]), "s"(base), "n"(c * 1024) : "memory");
                        } else {
                            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(voff[i]), "s"(base), "n"(c * 1024) : "memory");
                        }
                    }
                }
            } else {  // tail of the corpus (and look-ahead past it): clamp the rows; harmless re-reads keep vmcnt uniform
                char* dst0 = smem + (t & 1) * STAGE + 8 * lq * PITCH;
#pragma unroll
                for (int i = i0; i < i1; ++i) {
                    int32_t row = row0 + 8 * lq + i;
                    row = row < last_row ? row : last_row;
                    const char* p = src + (int64_t)row * ROWB + lane_off;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        if (!(HALF_TAIL && c == NCH - 1) || lane < TAIL_LANES)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + c * 1024),
                                                             (__attribute__((address_space(3))) void*)(dst0 + i * PITCH + c * 1024), 16, 0, /*nt*/ 2);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using IS = std::integral_constant<int, SPLR>;
        using I8 = std::integral_constant<int, 8>;
        // K-reduction of the previous tile's partials, done HERE because the loaders have to pause between B1 and B2
        // anyway (stage t&1 is still being read) and because in that window the compute waves issue LDS reads, not
        // MFMAs, so a wave sharing their SIMD is not starved of issue slots (measured: the same 60 instructions took
        // ~4.5 k cycles on an epilogue wave during the MFMA phase).  Loader l reduces row groups 2l and 2l+1 (one per
        // half wave) for every query column; C/D layout: lane (16g + j) of half h holds rows 4g..4g+3 of query 16h + j.
        auto kreduce = [&](int te) {
            const int qcL = lane & (NQC - 1), g = 2 * lq + ((lane >> 5) & 1);
            const int idx = 16 * g + (qcL & 15), qhL = qcL >> 4;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + qhL) * 64 + idx) * 16);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + ((1 * 2 + qhL) * 64 + idx) * 16);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(red + ((2 * 2 + qhL) * 64 + idx) * 16);
            const f32x4 p3 = *reinterpret_cast<const f32x4*>(red + ((3 * 2 + qhL) * 64 + idx) * 16);
            const f32x4 v = (p0 + p1) + (p2 + p3);  // fixed order: deterministic
            if constexpr (MODE == 0) {
                *reinterpret_cast<f32x4*>(ST + qcL * ST_PITCH + 4 * g) = v;  // transposed: one 16-B store per lane
            } else {
                const int32_t row0 = r_lo32 + te * TR;
                const int nvalid = ((int32_t)r_hi - row0) < TR ? ((int32_t)r_hi - row0) : TR;
                if ((lane & 31) < NQC && qcL < nq) {
                    float* o = out + (int64_t)qcL * ld + row0 + 4 * g;
                    if (4 * g + 3 < nvalid && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                        *reinterpret_cast<f32x4*>(o) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (4 * g + r < nvalid) o[r] = v[r];
                    }
                }
            }
        };
        // The DMA stream runs SPLR of a tile's 8 rows ahead of the barrier pair: when the loader stops for B1(t)/B2(t), the
        // first SPLR rows of tile t+1 are in flight and the rest follow right after B2(t).
        dma_rows(0, I0{}, I8{});
        dma_rows(1, I0{}, IS{});
        for (int t = 0; t < nt; ++t) {
            stamp(t, 0);
            // Loads retire in order: at most SPLR * NCH outstanding <=> tile t has landed (stores issued by kreduce in
            // between only make the wait stricter).
            if (ORD && lq == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPLR * NCH + 1) : "memory");  // + the ordinals of tile t+1
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPLR * NCH) : "memory");
            stamp(t, 1);
            wg_barrier();  // B1(t)
            stamp(t, 2);
            if (t > 0) kreduce(t - 1);
            wg_barrier();  // B2(t): stage t&1 is free, the reduced tile t-1 is in ST
            stamp(t, 3);
            dma_rows(t + 1, IS{}, I8{});  // stage (t+1)&1, free since B2(t-1)
            dma_rows(t + 2, I0{}, IS{});  // stage t&1, free since B2(t)
            stamp(t, 4);
        }
        wg_barrier();  // B1(nt)
        kreduce(nt - 1);
        wg_barrier();  // B2(nt)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
        dump_trace();
        return;
    }

    // ==================================== EPILOGUE WAVES (6, 7) =====================================================
    // MODE 1 is finished by the loaders' K-reduction; in MODE 0 wave 6 turns each reduced tile into per-chunk scores and
    // wave 7 only keeps the barrier count.  This wave shares a SIMD with a compute wave and gets an issue slot only
    // every ~60 cycles while that wave multiplies (measured), so whatever it does between B2 and the next B1 must be
    // SHORT or it delays every B1 (the first version's ~280 instructions did, by ~2 k cycles).  Hence:
    //   * the chunk ordinals of a tile's rows arrive in LDS with the tile (ring OFF_ORD); lane i compares the ordinals
    //     of rows i and i+1 and one v_cmp yields the chunk-end mask as a scalar;
    //   * every lane owns one query column and walks the 16 rows with ONE v_max per row (inline asm: fmaxf() would add
    //     two canonicalising v_max) and a wave-uniform branch where a chunk ends;
    //   * a chunk's sum over the query columns is a 4-step DPP butterfly (+ one row broadcast for 32 columns) done on
    //     the spot -- fixed order, deterministic -- and stored by one lane; the running maximum of the open chunk stays
    //     in a register across tiles;
    //   * the walk of tile t-1 runs in the NEXT B1..B2 window, where the compute waves read LDS instead of multiplying
    //     and nobody is starved (after B2(t) this wave only copies its ST column into registers);
    //   * finished chunks are collected lane-indexed (lane s = s-th chunk closed in the tile) and leave with ONE store
    //     per tile, issued after B2 where waiting behind the loaders' DMA backlog costs nothing.
    const int ew = wv - 6;
    const int qc = lane & (NQC - 1);
    const bool col_on = qc < nq;
    auto dpp_add = [](float x, auto CTRL) {
        return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(CTRL)::value, 0xf, 0xf, false));
    };
    float m = -INFINITY;  // running maximum of the chunk that is still open (per query column)
    float sv[TR];         // this lane's query column of the tile being walked
    float xacc = 0.f;     // lane s: score of the s-th chunk closed in the tile ...
    int32_t cacc = 0;     // ... and its ordinal
    int nclosed = 0;      // wave-uniform
#pragma unroll
    for (int i = 0; i < TR; ++i) sv[i] = -INFINITY;
    int32_t rcl = 0;    // lane i: chunk ordinal of row i of the tile being walked
    uint32_t ends = 0;  // bit i: row i is the last row of its chunk (wave-uniform)
    // Chunk-end mask of tile te (its ordinals landed with the tile, before B1(te)): lane i compares rows i and i+1.
    auto fetch_ordinals = [&](int te) {
        const int32_t* ord = reinterpret_cast<const int32_t*>(smem + OFF_ORD + (te & 3) * 256);
        rcl = ord[lane];
        const int32_t rcn = ord[lane + 1 < 64 ? lane + 1 : 63];
        ends = (uint32_t)__builtin_amdgcn_ballot_w64(rcl != rcn) & 0xffffu;
        const int32_t left = (int32_t)r_hi - (r_lo32 + te * TR);
        if (left < TR) ends &= (1u << left) - 1u;  // rows past this workgroup's range belong to its neighbour
    };
    // Rows of the fetched tile: segmented running max; a closed chunk is summed over the query columns at once.
    auto walk = [&]() {
        nclosed = 0;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(sv[i]));
            if (__builtin_expect((ends >> i) & 1u, 0)) {
                float x = col_on ? m : 0.f;  // padded query columns add 0
                x = dpp_add(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
                x = dpp_add(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
                x = dpp_add(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
                x = dpp_add(x, std::integral_constant<int, 0x140>{});  // row_mirror: every lane = sum of its 16
                if constexpr (NQT == 2)  // lanes 16..31 += lane 15 (columns 0..15)
                    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));
                const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16 * (NQT - 1)));
                const int32_t cid = __builtin_amdgcn_readlane(rcl, i);
                xacc = lane == nclosed ? total : xacc;
                cacc = lane == nclosed ? cid : cacc;
                ++nclosed;
                m = -INFINITY;
            }
        }
    };
    const bool worker = MODE == 0 && ew == 0;  // wave-uniform
    // MODE 0 reuses `ld` (meaningless without a score matrix) as a flag: add this pass's chunk scores to `out` instead
    // of overwriting them.  That is how more than 32 query vectors are handled: 32 at a time, summed in pass order
    // (every chunk is owned by exactly one lane of one workgroup, so the read-modify-write needs no atomics).
    const bool accumulate = MODE == 0 && ld != 0;
    for (int t = 0; t <= nt; ++t) {  // iteration t: walk tile t-2 in the window, fetch tile t-1 after it; t = nt drains
        stamp(t, 0);
        wg_barrier();  // B1(t)
        stamp(t, 1);
        if (worker && t > 1) walk();  // tile t-2: pure VALU/SALU work, nothing that queues behind the A-fragment reads
        stamp(t, 2);
        wg_barrier();  // B2(t): the loaders have put the reduced tile t-1 into ST
        stamp(t, 3);
        if (worker && t > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ST + qc * ST_PITCH + 4 * g);
                sv[4 * g + 0] = v[0]; sv[4 * g + 1] = v[1]; sv[4 * g + 2] = v[2]; sv[4 * g + 3] = v[3];
            }
            const int32_t left = (int32_t)r_hi - (r_lo32 + (t - 1) * TR);
            if (left < TR) {  // last tile of the range only
#pragma unroll
                for (int i = 0; i < TR; ++i) sv[i] = i < left ? sv[i] : -INFINITY;
            }
            if (t > 1 && lane < nclosed) out[cacc] = accumulate ? out[cacc] + xacc : xacc;  // tile t-2's chunks
            fetch_ordinals(t - 1);
            stamp(t, 4);
        }
    }
    if (worker) {
        walk();
        if (lane < nclosed) out[cacc] = accumulate ? out[cacc] + xacc : xacc;
    }
    dump_trace();
}

__global__ __launch_bounds__(256) void row_to_chunk_kernel(const int64_t* __restrict__ chunk_offsets,
                                                            int64_t n_chunks, int64_t n_rows,
                                                            int32_t* __restrict__ row_to_chunk) {
    // One thread per chunk writes its rows' ordinals; rc[n_rows .. n_rows+64] = -1 terminates the last chunk and
    // pads the array so that a tile's 64-entry ordinal DMA (first row clamped to the corpus) never leaves it.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const int64_t b = chunk_offsets[c], e = chunk_offsets[c + 1];
        for (int64_t r = b; r < e; ++r) row_to_chunk[r] = (int32_t)c;
    }
    if (blockIdx.x == 0 && threadIdx.x < 65) row_to_chunk[n_rows + threadIdx.x] = -1;
}

int launch_row_to_chunk(const int64_t* chunk_offsets, int64_t n_chunks, int64_t n_rows, int32_t* row_to_chunk,
                        hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 255) / 256, 4096));
    hipLaunchKernelGGL(row_to_chunk_kernel, dim3(blocks), dim3(256), 0, s, chunk_offsets, n_chunks, n_rows,
                       row_to_chunk);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// =====================================================================================================================
// TWO QUERIES PER CORPUS PASS (batched MaxSim, SPLIT arithmetic only).
//
// With the fp16 (hi, lo) split a tile costs a compute wave 48 MFMAs of 16 cycles + ~160 conversion VALU ops -- a third
// of the ~4.7 k cycles the tile's 64 KiB take to arrive -- so the kernel above is HBM-bound and the matrix pipe idles.
// What stops it from scoring more query vectors per pass is the register file: a K-quarter of 32 query vectors as
// (hi, lo) pairs is 128 VGPRs of a wave's 256.  This variant therefore makes all eight waves symmetric: wave
// (g, w) = (wv >> 2, wv & 3) multiplies K-quarter w of the tile with query g's 32 vectors (both groups read the same
// A fragments from LDS), and the loader / K-reduce / epilogue duties of the kernel above are spread over them:
//   * every wave brings 2 rows of each tile (LDS-DMA, hand-issued as above; tile t+2 right after B2(t)); wave 0 also
//     brings the tile's chunk ordinals;
//   * waves w = 0, 1 of a group K-reduce that group's partials, wave w = 3 walks its 32 query columns (segmented max,
//     DPP column sum) exactly as wave 6 does above;
//   * same two barriers per tile, same data flow between them (partials -> ST -> walker registers -> one store).
// One launch = one corpus pass = two queries' chunk scores: out[g * out_stride + chunk].
template <int KW, bool F16 = false>
struct Geo2 {
    using G1 = Geo<KW, F16>;
    static constexpr int OFF_RED = NSTAGE * G1::STAGE;              // 2 groups x 8 KiB of K-partials
    static constexpr int OFF_ST = OFF_RED + 2 * G1::RED_BYTES;
    static constexpr int ST_BYTES = 32 * G1::ST_PITCH * 4;          // per group and tile parity: [32 query columns][20] fp32
    static constexpr int OFF_ORD = OFF_ST + 4 * ST_BYTES;           // (two tiles deep: the walker reads tile t-2 while
    static constexpr int LDS_TOTAL = OFF_ORD + 4 * 256;             //  tile t-1 is being reduced)  159 744 B at KW = 256
};

// Query side of the SPLIT arithmetic, done ONCE per query instead of in every corpus pass's prologue (where its ~1.5 k
// instructions per wave cost ~12 us -- 13 % of a pass over a 125 k-row shard): wave (query, w) scales its K-quarter of the
// query's vectors to [2^13, 2^14), splits every element into fp16 (hi, lo) and writes the MFMA B fragments in exactly
// the register layout the pass kernels hold them in:
//   frag[((((query * 4 + w) * 2 + h) * MS + m) * 2 + part) * 64 + lane]   (16 B: 8 fp16; part 0 = hi, 1 = lo)
//   meta[(query * 4 + w) * 2 + {0, 1}] = {2^(ex - 14) (undoes the scale), any lo != 0}
template <int KW, bool F16>
__global__ __launch_bounds__(64) void query_split_kernel(const float* __restrict__ Q, int nq, int64_t q_stride,
                                                          uint4* __restrict__ frag, float* __restrict__ meta) {
    constexpr int SD = 4 * KW, MS = KW / 32, NQT = 2;
    const int lane = threadIdx.x, w = blockIdx.x;
    const int64_t query = blockIdx.y;
    const float* const Qg = Q + query * q_stride;
    const int fj = lane & 15, kq = lane >> 4;
    float mx = 0.f;
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
        const int qi = 16 * h + fj;
        if (qi < nq)
            for (int c = 0; c < KW / 4; c += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(Qg + (int64_t)qi * SD + KW * w + KW / 4 * kq + c);
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    int ex = 0;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
    const float q_scale = ldexpf(1.f, 14 - ex);
    bool any_lo = false;
    uint4* const out = frag + (query * 4 + w) * (NQT * MS * 2 * 64) + lane;
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
        const int qi = 16 * h + fj;
        const int qc_ = qi < nq ? qi : nq - 1;  // clamped load, zeroed below: padded query vectors add 0
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            // fp32 corpus: k = 16 (2m + (u >> 2)) + 4 kq + (u & 3); fp16-stored corpus: k = 32 m + 8 kq + u
            const float* qp = Qg + (int64_t)qc_ * SD + KW * w + 32 * m + (F16 ? 8 : 4) * kq;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp), v1 = *reinterpret_cast<const f32x4*>(qp + (F16 ? 4 : 16));
            h16x8 hi8, lo8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = qi < nq ? (u < 4 ? v0[u] : v1[u - 4]) * q_scale : 0.f;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)(x - (float)hi);
                hi8[u] = hi;
                lo8[u] = lo;
                any_lo |= lo != (_Float16)0.0f;
            }
            uint4 a, b;
            __builtin_memcpy(&a, &hi8, 16);
            __builtin_memcpy(&b, &lo8, 16);
            out[((h * MS + m) * 2 + 0) * 64] = a;
            out[((h * MS + m) * 2 + 1) * 64] = b;
        }
    }
    any_lo = __builtin_amdgcn_ballot_w64(any_lo) != 0;
    if (lane == 0) {
        meta[(query * 4 + w) * 2 + 0] = ldexpf(1.f, ex - 14);
        meta[(query * 4 + w) * 2 + 1] = any_lo ? 1.f : 0.f;
    }
}

template <int KW, bool TRACE = false, bool F16 = false>
__global__ __launch_bounds__(512, 2) void maxsim_stream2_kernel(const float* __restrict__ D, int64_t n_rows,
                                                                  const uint4* __restrict__ qfrag,
                                                                  const float* __restrict__ qmeta, int nq,
                                                                  const int32_t* __restrict__ row_to_chunk,
                                                                  const int64_t* __restrict__ chunk_offsets,
                                                                  int64_t n_chunks, float* __restrict__ out,
                                                                  int64_t out_stride, float e_scale,
                                                                  unsigned long long* trace) {
    using G1 = Geo<KW, F16>;
    using G2 = Geo2<KW, F16>;
    constexpr int PITCH = G1::PITCH, STAGE = G1::STAGE, NCH = G1::NCH, ROWB = G1::ROWB, ST_PITCH = G1::ST_PITCH;
    constexpr int KSTEPS = G1::KSTEPS, MS = KW / 32, NQT = 2, NQC = 32;  // fragment reads per tile; fp16 MFMA steps
    __shared__ __attribute__((aligned(16))) char smem[G2::LDS_TOTAL + (TRACE ? 4096 : 0)];
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();  // 0..7
    // Tile timeline of workgroup 7, tiles 100..107 (diagnostic build, RAGLITE_HIP_TRACE2=1): s_memtime stamps kept in LDS.
    auto stamp = [&](int t, int k) {
        if constexpr (TRACE) {
            if (blockIdx.x == 7 && t >= 100 && t < 108 && lane == 0)
                reinterpret_cast<unsigned long long*>(smem + G2::LDS_TOTAL)[((t - 100) * 8 + wv) * 8 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    if constexpr (TRACE) {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) reinterpret_cast<unsigned long long*>(smem + G2::LDS_TOTAL)[i] = 0;
        __syncthreads();
    }
    const int w = wv & 3, g = wv >> 2;
    const int64_t G = gridDim.x, b = blockIdx.x;
    auto boundary = [&](int64_t t) -> int64_t {
        if (t <= 0) return 0;
        if (t >= n_rows) return n_rows;
        const int32_t c = row_to_chunk[t];
        const int64_t c0 = chunk_offsets[c], c1 = chunk_offsets[c + 1];
        return c0 == t ? t : c1;
    };
    const int64_t r_lo = uniform_i64(boundary((n_rows * b) / G));
    const int64_t r_hi = uniform_i64((b + 1 == G) ? n_rows : boundary((n_rows * (b + 1)) / G));
    const int nt = (int)((r_hi - r_lo + TR - 1) / TR);
    if (nt <= 0) return;
    const int32_t last_row = (int32_t)(n_rows - 1);
    const int32_t r_lo32 = (int32_t)r_lo;
    char* const red = smem + G2::OFF_RED + g * G1::RED_BYTES;
    float* const ST0 = reinterpret_cast<float*>(smem + G2::OFF_ST + 2 * g * G2::ST_BYTES);  // [tile parity][32][20]

    // ---- this wave's slice of its query as (hi, lo) fp16 MFMA B fragments, prepared by query_split_kernel ------------------
    const int fj = lane & 15, kq = lane >> 4;
    h16x8 qhi[NQT][MS], qlo[NQT][MS];
    {
        const uint4* const fr = qfrag + (int64_t)(g * 4 + w) * (NQT * MS * 2 * 64) + lane;
#pragma unroll
        for (int h = 0; h < NQT; ++h)
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const uint4 x = fr[((h * MS + m) * 2 + 0) * 64], y = fr[((h * MS + m) * 2 + 1) * 64];
                __builtin_memcpy(&qhi[h][m], &x, 16);
                __builtin_memcpy(&qlo[h][m], &y, 16);
            }
    }
    const float q_unscale = F16 ? qmeta[(g * 4 + w) * 2 + 0] : qmeta[(g * 4 + w) * 2 + 0] / e_scale;  // stored halves are not rescaled
    const bool any_lo = __builtin_amdgcn_readfirstlane(__float_as_int(qmeta[(g * 4 + w) * 2 + 1])) != 0;
    const char* const a_base = smem + fj * PITCH + w * G1::QBYTES + kq * 16;

    // ---- this wave's share of the DMA stream: rows 2 wv, 2 wv + 1 of every tile ------------------------------------------
    const char* const src = reinterpret_cast<const char*>(D);
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lds0 = lds_base + (uint32_t)(2 * wv * PITCH);
    const uint32_t lds_ord = lds_base + G2::OFF_ORD;
    const uint32_t voff_ord = 4u * lane;
    constexpr int TAIL_LANES = (ROWB % 1024) / 16;
    constexpr bool HALF_TAIL = TAIL_LANES != 0;
    constexpr int DMA_PER_TILE = 2 * NCH;
    // One tile = 2 * NCH DMA instructions per wave (+ the ordinals, wave 0).  They are issued ONE AT A TIME from inside the
    // MFMA loop of the tile two behind: all eight waves firing their DMAs at once right after B2 costs each of them
    // 0.3-1.6 k cycles of issue stall on the critical path (measured, profiles/r01_i_pair_timeline.txt).
    struct TileSrc { const char* row[2]; uint32_t lds; };
    auto tile_src = [&](int t) {
        TileSrc ts;
        const int32_t row0 = r_lo32 + t * TR + 2 * wv;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int32_t row = row0 + i;
            row = row < last_row ? row : last_row;  // past the corpus: harmless re-reads of the last row, never used
            ts.row[i] = src + (int64_t)row * ROWB;
        }
        ts.lds = lds0 + (uint32_t)((t & 1) * STAGE);
        return ts;
    };
    auto dma_ordinals = [&](int t) {
        if (wv == 0) {  // wave-uniform: the tile's 64 chunk ordinals, from its first row on
            const int32_t row0 = r_lo32 + t * TR;
            const int32_t rowc = row0 < last_row ? row0 : last_row;
            const int32_t* rc = row_to_chunk + rowc;
            const uint32_t dst = lds_ord + (uint32_t)((t & 3) * 256);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(dst), "v"(voff_ord), "s"(rc)
                         : "memory", "m0");
        }
    };
    auto dma_one = [&](const TileSrc& ts, auto J_) {  // DMA j = NCH * i + c of the tile: 1 KiB (or the half tail) of row i
        constexpr int j = decltype(J_)::value, i = j / NCH, c = j % NCH;
        const uint32_t lane_off = 16u * lane;  // (asm operands alone do not capture an enclosing local in a generic lambda)
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0" ::"s"(ts.lds), "n"(i * PITCH) : "memory", "m0", "scc");
        if (HALF_TAIL && c == NCH - 1) {
            if (lane < TAIL_LANES)
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(lane_off), "s"(ts.row[i]), "n"(c * 1024) : "memory");
        } else {
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2 nt" ::"v"(lane_off), "s"(ts.row[i]), "n"(c * 1024) : "memory");
        }
    };
    auto dma_tile = [&](int t) {  // the whole tile at once (prologue only)
        dma_ordinals(t);
        const TileSrc ts = tile_src(t);
        [&]<int... J>(std::integer_sequence<int, J...>) { (dma_one(ts, std::integral_constant<int, J>{}), ...); }
        (std::make_integer_sequence<int, DMA_PER_TILE>{});
    };

    // ---- K-reduction of a tile's partials (waves w = 0, 1 of the group), as the loaders do above ----------------------
    auto kreduce = [&](int te) {
        float* const ST = ST0 + (te & 1) * (G2::ST_BYTES / 4);
        const int qcL = lane & (NQC - 1), gg = 2 * w + ((lane >> 5) & 1);
        const int idx = 16 * gg + (qcL & 15), qhL = qcL >> 4;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + qhL) * 64 + idx) * 16);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + ((1 * 2 + qhL) * 64 + idx) * 16);
        const f32x4 p2 = *reinterpret_cast<const f32x4*>(red + ((2 * 2 + qhL) * 64 + idx) * 16);
        const f32x4 p3 = *reinterpret_cast<const f32x4*>(red + ((3 * 2 + qhL) * 64 + idx) * 16);
        *reinterpret_cast<f32x4*>(ST + qcL * ST_PITCH + 4 * gg) = (p0 + p1) + (p2 + p3);
    };

    // ---- the walkers: the epilogue wave of the kernel above, one per query group ------------------------------------------
    // The walk of a tile takes ~1.3 k cycles of dependent code.  The older waves of each SIMD (0-3, group 0) get through
    // their MFMA phase ~0.8 k cycles before the younger ones (measured), so the two walkers are waves 3 (group 0's
    // columns) and 2 (group 1's), and they walk tile t-1 AFTER their MFMA phase of tile t, in time they would otherwise
    // spend waiting at the next barrier -- not in the B1..B2 window, where every other wave would wait for them.
    const bool walker = wv == 3 || wv == 2;   // wave-uniform
    const int wg_ = wv == 3 ? 0 : 1;          // whose columns this walker owns
    const bool reducer = w < 2;
    const float* const STw = reinterpret_cast<const float*>(smem + G2::OFF_ST + 2 * wg_ * G2::ST_BYTES);
    float* const outw = out + (int64_t)wg_ * out_stride;

    const int qc = lane & (NQC - 1);
    const bool col_on = qc < nq;
    auto dpp_add = [](float x, auto CTRL) {
        return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(CTRL)::value, 0xf, 0xf, false));
    };
    float m = -INFINITY;
    float xacc = 0.f;
    int32_t cacc = 0;
    int nclosed = 0;
    int32_t rcl = 0;
    uint32_t ends = 0;
    auto fetch_ordinals = [&](int te) {
        const int32_t* ord = reinterpret_cast<const int32_t*>(smem + G2::OFF_ORD + (te & 3) * 256);
        rcl = ord[lane];
        const int32_t rcn = ord[lane + 1 < 64 ? lane + 1 : 63];
        ends = (uint32_t)__builtin_amdgcn_ballot_w64(rcl != rcn) & 0xffffu;
        const int32_t left = (int32_t)r_hi - (r_lo32 + te * TR);
        if (left < TR) ends &= (1u << left) - 1u;
    };
    auto walk = [&](int te) {  // this lane's query column of the reduced tile te, straight from LDS
        const float* const ST = STw + (te & 1) * (G2::ST_BYTES / 4);
        float sv[TR];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ST + qc * ST_PITCH + 4 * q4);
            sv[4 * q4 + 0] = v[0]; sv[4 * q4 + 1] = v[1]; sv[4 * q4 + 2] = v[2]; sv[4 * q4 + 3] = v[3];
        }
        const int32_t left = (int32_t)r_hi - (r_lo32 + te * TR);
        if (left < TR) {  // last tile of the range only
#pragma unroll
            for (int i = 0; i < TR; ++i) sv[i] = i < left ? sv[i] : -INFINITY;
        }
        nclosed = 0;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(sv[i]));
            if (__builtin_expect((ends >> i) & 1u, 0)) {
                float x = col_on ? m : 0.f;
                x = dpp_add(x, std::integral_constant<int, 0xB1>{});
                x = dpp_add(x, std::integral_constant<int, 0x4E>{});
                x = dpp_add(x, std::integral_constant<int, 0x141>{});
                x = dpp_add(x, std::integral_constant<int, 0x140>{});
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));
                const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
                const int32_t cid = __builtin_amdgcn_readlane(rcl, i);
                xacc = lane == nclosed ? total : xacc;
                cacc = lane == nclosed ? cid : cacc;
                ++nclosed;
                m = -INFINITY;
            }
        }
    };
    dma_tile(0);
    if (nt > 1) dma_tile(1);
    for (int t = 0; t <= nt; ++t) {
        const bool live = t < nt;  // iteration nt only drains the epilogue pipeline
        stamp(t, 0);
        if (live) {
            // Loads retire in order: when at most one tile's worth of this wave's DMAs is outstanding, tile t has landed
            // (the walker's stores in between only make the wait stricter).
            if (t + 1 < nt) {
                if (wv == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE + 1) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        stamp(t, 1);
        wg_barrier();  // B1(t): tile t is in stage t&1; the K-partials of tile t-1 are in LDS
        stamp(t, 2);
        f32x4 a[KSTEPS];
        if (live) {
            const char* ap = a_base + (t & 1) * STAGE;
#pragma unroll
            for (int mm = 0; mm < KSTEPS; ++mm) a[mm] = *reinterpret_cast<const f32x4*>(ap + mm * 64);
        }
        if (reducer && t > 0) kreduce(t - 1);
        stamp(t, 3);
        wg_barrier();  // B2(t): every wave holds tile t in registers; the reduced tile t-1 is in ST
        stamp(t, 4);
        const bool feed = t + 2 < nt;                    // tile t+2 goes into stage t&1, free since this barrier
        TileSrc ts{};
        if (feed) { ts = tile_src(t + 2); dma_ordinals(t + 2); }
        stamp(t, 5);
        if (live) {
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[NQT], acl[NQT];
#pragma unroll
            for (int h = 0; h < NQT; ++h) { acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; acl[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int mi = 0; mi < MS; ++mi) {
                if constexpr (F16) {  // the stored halves ARE the operand: two MFMAs per query tile, no conversion
                    h16x8 af;
                    __builtin_memcpy(&af, &a[mi], 16);
#pragma unroll
                    for (int h = 0; h < NQT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qhi[h][mi], acc[h], 0, 0, 0);
                    if (any_lo) {
#pragma unroll
                        for (int h = 0; h < NQT; ++h) acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, qlo[h][mi], acl[h], 0, 0, 0);
                    }
                } else {
                    h16x8 eh, el;
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        // (two scalar multiplies, not one v_pk_mul_f32: packed fp32 VALU beside MFMAs costs ~+25 cycles each)
                        const float x0 = a[2 * mi + (u >> 2)][u & 3] * e_scale, x1 = a[2 * mi + (u >> 2)][(u & 3) + 1] * e_scale;
                        const auto ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);
                        const auto pl = __builtin_amdgcn_cvt_pkrtz(x0 - (float)ph[0], x1 - (float)ph[1]);
                        eh[u] = ph[0]; eh[u + 1] = ph[1];
                        el[u] = pl[0]; el[u + 1] = pl[1];
                    }
#pragma unroll
                    for (int h = 0; h < NQT; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, qhi[h][mi], acc[h], 0, 0, 0);
#pragma unroll
                    for (int h = 0; h < NQT; ++h) acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el, qhi[h][mi], acl[h], 0, 0, 0);
                    if (any_lo) {
#pragma unroll
                        for (int h = 0; h < NQT; ++h) acl[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh, qlo[h][mi], acl[h], 0, 0, 0);
                    }
                }
                if (feed) {  // this step's share of the DMA stream
                    [&]<int... J>(std::integer_sequence<int, J...>) {
                        ((J * MS / DMA_PER_TILE == mi ? dma_one(ts, std::integral_constant<int, J>{}) : (void)0), ...);
                    }(std::make_integer_sequence<int, DMA_PER_TILE>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 0; h < NQT; ++h)
                *reinterpret_cast<f32x4*>(red + ((w * 2 + h) * 64 + lane) * 16) = (acc[h] + acl[h]) * q_unscale;
        }
        stamp(t, 6);
        if (walker && t > 0) {  // tile t-1: reduced in this iteration's window, complete since B2(t)
            fetch_ordinals(t - 1);
            walk(t - 1);
            if (lane < nclosed) outw[cacc] = xacc;
        }
        stamp(t, 7);
    }
    if constexpr (TRACE) {
        if (blockIdx.x == 7) {
            const int i = ((lane >> 3) * 8 + wv) * 8 + (lane & 7);
            trace[i] = reinterpret_cast<unsigned long long*>(smem + G2::LDS_TOTAL)[i];
        }
    }
}

// Range of the corpus' row magnitudes, for the SPLIT arithmetic of the stream kernel: range[0] = largest |element|,
// range[1] = smallest row maximum over the rows that are not all zero, range[2] != 0 if an element is not finite.
// Magnitudes travel as uint32 bit patterns: non-negative floats order like their bits, and inf / NaN sort above every
// finite value.  (Maxima, not norms: a sum of squares underflows for corpora scaled far down.)  One wave per row.
__global__ __launch_bounds__(256) void row_range_kernel(const float* __restrict__ E, int64_t n_rows, int32_t dim,
                                                         uint32_t* __restrict__ range) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    uint32_t mx = 0u, mn = 0x7f800000u;
    for (int64_t r = wave; r < n_rows; r += nw) {
        const float* row = E + r * dim;
        uint32_t m = 0u;
        for (int c = 4 * lane; c < dim; c += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t bits = __float_as_uint(v[u]) & 0x7fffffffu;
                m = bits > m ? bits : m;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)m, o);
            m = other > m ? other : m;
        }
        mx = m > mx ? m : mx;
        if (m != 0u) mn = m < mn ? m : mn;
    }
    if (lane == 0) {
        atomicMax(range + 0, mx);
        atomicMin(range + 1, mn);
        if (mx >= 0x7f800000u) atomicOr(range + 2, 1u);
    }
}

int launch_row_range(const float* E, int64_t n_rows, int32_t dim, uint32_t* range, hipStream_t s) {
    if (dim % 4) return RL_ERR_UNSUPPORTED;
    const uint32_t init[3] = {0u, 0x7f800000u, 0u};
    RL_HIP(hipMemcpyAsync(range, init, sizeof(init), hipMemcpyHostToDevice, s));
    if (n_rows > 0) {
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_rows + 3) / 4, 256 * 8));
        hipLaunchKernelGGL(row_range_kernel, dim3(blocks), dim3(256), 0, s, E, n_rows, dim, range);
        RL_HIP(hipGetLastError());
    }
    return RL_OK;
}

namespace {
struct StreamArgs {
    const float* D; int64_t n_rows; const float* Q; int nq; const int32_t* r2c; const int64_t* off; int64_t n_chunks;
    int mode; float* out; int64_t ld; dim3 grid; hipStream_t s; unsigned long long* trace; float e_scale;
    int64_t q_batch_stride = 0, out_batch_stride = 0;
    StreamSecondJob j2 = {};
};
template <int KW, bool F16, bool SPLIT = false>
void launch_kw(const StreamArgs& a) {
    const dim3 blk(512);
#define RL_STREAM(NQT, MODE) hipLaunchKernelGGL((maxsim_stream_kernel<KW, NQT, MODE, false, 6, F16, SPLIT>), a.grid, blk, 0, a.s, a.D, \
                                                a.n_rows, a.Q, a.nq, a.r2c, a.off, a.n_chunks, a.out, a.ld, a.trace, a.e_scale, a.q_batch_stride, \
                                                a.out_batch_stride, a.j2)
    if (a.mode == 0) { if (a.nq <= 16) RL_STREAM(1, 0); else RL_STREAM(2, 0); }
    else             { if (a.nq <= 16) RL_STREAM(1, 1); else RL_STREAM(2, 1); }
#undef RL_STREAM
}
}  // namespace

// Fast path for dim in {128, 256, 384, 512, 768, 1024} (dim = 4 * KW, KW a multiple of 32) and nq <= 32.
// f16 = the corpus is stored as IEEE fp16 (D then points at uint16_t data); queries and scores stay fp32.
static int launch_stream_any(const float* D, bool f16, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                             const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                             float* out, int64_t ld, int n_cu, hipStream_t s, float split_scale = 0.f, const uint32_t* run_if = nullptr,
                             int32_t n_batch = 1, int64_t q_batch_stride = 0, int64_t out_batch_stride = 0, const StreamSecondJob* job2 = nullptr) {
    if (nq < 1 || nq > 32 || n_rows < 1 || n_batch < 1 || n_batch > 65535 || (n_batch > 1 && mode != 0)) return RL_ERR_UNSUPPORTED;
    if (job2 && (n_batch != 1 || !job2->D || job2->n_rows < 1 || (reinterpret_cast<uintptr_t>(job2->D) & 15))) return RL_ERR_UNSUPPORTED;
    if (dim != 128 && dim != 256 && dim != 384 && dim != 512 && dim != 768 && dim != 1024) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t tiles = (std::max<int64_t>(n_rows, job2 ? job2->n_rows : 0) + TR - 1) / TR;  // (workgroups without a tile of their job return)
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles));
    if (job2) { n_batch = 2; q_batch_stride = out_batch_stride = 0; }  // grid row 1 = the second job (same queries)
#ifdef RAGLITE_EXPERIMENTS  // the tile-timeline build of the kernel exists in experiment builds only
    static unsigned long long* trace = [] {
        unsigned long long* p = nullptr;
        if (exp_env("RAGLITE_HIP_TRACE")) { (void)hipMalloc(&p, 8 * 8 * 8 * 8); (void)hipMemset(p, 0, 8 * 8 * 8 * 8); }
        return p;
    }();
    if (trace && !f16 && mode == 0 && nq > 16 && dim == 1024) {  // diagnostic build: dump the 30th launch's timeline to stderr
        static int calls = 0;
        hipLaunchKernelGGL((maxsim_stream_kernel<256, 2, 0, true>), dim3(grid), dim3(512), 0, s, D, n_rows, Q, nq,
                           row_to_chunk, chunk_offsets, n_chunks, out, ld, trace, 1.f, (int64_t)0, (int64_t)0, StreamSecondJob{});
        if (++calls == 30) {
            unsigned long long h[8 * 8 * 8];
            (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
            const unsigned long long t0 = h[6 * 64];  // first stamp of the workgroup's first tile
            fprintf(stderr, "TRACE columns: compute 0-3: arrive-B1 after-B1 after-A-reads after-B2 after-MFMA | "
                            "loader 4-5: before-vmcnt after-vmcnt after-B1 after-B2 after-DMA-issue | epilogue 6-7: "
                            "arrive-B1 after-B1 after-walk(t-2) after-B2 after-ST-copy/store/ordinals(t-1)   (s_memtime ticks "
                            "since the first stamp; rows 'tile 106/107' are the workgroup's FIRST and LAST tile)\n");
            for (int t = 0; t < 8; ++t)
                for (int wv = 0; wv < 8; ++wv) {
                    fprintf(stderr, "TRACE tile %d wave %d:", t + 100, wv);
                    for (int k = 0; k < 8; ++k) {
                        const unsigned long long v = h[(t * 8 + wv) * 8 + k];
                        if (v) fprintf(stderr, " %8lld", (long long)(v - t0)); else fprintf(stderr, "        -");
                    }
                    fprintf(stderr, "\n");
                }
        }
        return RL_OK;
    }
#endif
    const StreamArgs a{D, n_rows, Q, (int)nq, row_to_chunk, chunk_offsets, n_chunks, mode, out, ld, dim3(grid, (unsigned)n_batch), s,
                       reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(run_if)), split_scale, q_batch_stride, out_batch_stride,
                       job2 ? *job2 : StreamSecondJob{}};
    const bool split = !f16 && split_scale > 0.f;  // 0: the exact fp32 MFMA chain
#define RL_DIMS(...) switch (dim) { \
        case 128: launch_kw<32, __VA_ARGS__>(a); break; case 256: launch_kw<64, __VA_ARGS__>(a); break; case 384: launch_kw<96, __VA_ARGS__>(a); break; \
        case 512: launch_kw<128, __VA_ARGS__>(a); break; case 768: launch_kw<192, __VA_ARGS__>(a); break; default: launch_kw<256, __VA_ARGS__>(a); break; }
    if (f16) { RL_DIMS(true) } else if (split) { RL_DIMS(false, true) } else { RL_DIMS(false) }
#undef RL_DIMS
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_maxsim_stream(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                         const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                         float* out, int64_t ld, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if) {
    return launch_stream_any(D, false, n_rows, dim, Q, nq, row_to_chunk, chunk_offsets, n_chunks, mode, out, ld, n_cu, s,
                             split_scale, run_if);
}

// launch_maxsim_stream with a SECOND job in the same launch (grid row 1): the same queries over job2's matrix into job2's output, behind
// job2's run-if flag.  Mode 1 (row scores) only; fp32 rows.
int launch_maxsim_stream_two(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq, const int32_t* row_to_chunk,
                             const int64_t* chunk_offsets, int64_t n_chunks, float* out, int64_t ld, int n_cu, hipStream_t s, float split_scale,
                             const StreamSecondJob& job2) {
    return launch_stream_any(D, false, n_rows, dim, Q, nq, row_to_chunk, chunk_offsets, n_chunks, 1, out, ld, n_cu, s, split_scale, nullptr, 1, 0,
                             0, &job2);
}

// MaxSim chunk scores of a BATCH of queries in one launch (grid row y = query y: Q + y q_stride -> out + y out_stride), fp32 or fp16 rows;
// what the guarded fallback of a MaxSim batch runs over an index without a pre-split image.  nq <= 32 per query.
int launch_maxsim_stream_batch(const void* D, bool f16, int64_t n_rows, int32_t dim, const float* Q, int32_t nq, int64_t q_stride,
                               int32_t n_queries, const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, float* out,
                               int64_t out_stride, int n_cu, hipStream_t s, float split_scale, const uint32_t* run_if) {
    return launch_stream_any(static_cast<const float*>(D), f16, n_rows, dim, Q, nq, row_to_chunk, chunk_offsets, n_chunks, 0, out, 0, n_cu, s,
                             f16 ? 0.f : split_scale, run_if, n_queries, q_stride, out_stride);
}

// Two queries (17..32 vectors each, q_stride floats apart) per corpus pass over an fp32 corpus in SPLIT arithmetic:
// out[g * out_stride + chunk], g = 0, 1.  RL_ERR_UNSUPPORTED outside that shape (the caller then makes two passes).
// SPLIT query fragments for `n_queries` queries of nq (17..32) vectors, q_stride floats apart (see query_split_kernel).
size_t query_split_bytes(int32_t dim, int32_t n_queries) { return (size_t)n_queries * (4 * 2 * (dim / 128) * 2 * 64 * 16 + 64); }
int launch_query_split(const float* Q, int32_t dim, int32_t nq, int64_t q_stride, int32_t n_queries, void* buf, bool f16_corpus,
                       hipStream_t s) {
    if (nq <= 16 || nq > 32 || n_queries < 1) return RL_ERR_UNSUPPORTED;
    if (dim != 128 && dim != 256 && dim != 384 && dim != 512 && dim != 768 && dim != 1024) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(Q) & 15) || (q_stride & 3)) return RL_ERR_UNSUPPORTED;
    uint4* frag = static_cast<uint4*>(buf);
    float* meta = reinterpret_cast<float*>(static_cast<char*>(buf) + (size_t)n_queries * (4 * 2 * (dim / 128) * 2 * 64 * 16));
    const dim3 grid(4, (unsigned)n_queries), blk(64);
#define RL_QS(KW) do { if (f16_corpus) hipLaunchKernelGGL((query_split_kernel<KW, true>), grid, blk, 0, s, Q, (int)nq, q_stride, frag, meta); \
                       else hipLaunchKernelGGL((query_split_kernel<KW, false>), grid, blk, 0, s, Q, (int)nq, q_stride, frag, meta); } while (0)
    switch (dim) {
        case 128: RL_QS(32); break; case 256: RL_QS(64); break; case 384: RL_QS(96); break;
        case 512: RL_QS(128); break; case 768: RL_QS(192); break; default: RL_QS(256); break;
    }
#undef RL_QS
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Two queries (17..32 vectors each; queries `first`, `first + 1` of a launch_query_split buffer over `n_queries`) per corpus
// pass over an fp32 corpus in SPLIT arithmetic: out[g * out_stride + chunk], g = 0, 1.  RL_ERR_UNSUPPORTED outside that
// shape (the caller then makes one pass per query).
int launch_maxsim_stream2(const void* Dv, bool f16, int64_t n_rows, int32_t dim, const void* split_buf, int32_t n_queries,
                          int32_t first, int32_t nq, const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks,
                          float* out, int64_t out_stride, int n_cu, hipStream_t s, float split_scale) {
    const float* D = static_cast<const float*>(Dv);  // fp32 rows, or (f16) IEEE halves
    if (nq <= 16 || nq > 32 || n_rows < 1 || (!f16 && !(split_scale > 0.f)) || first < 0 || first + 2 > n_queries) return RL_ERR_UNSUPPORTED;
    if (dim != 128 && dim != 256 && dim != 384 && dim != 512 && dim != 768 && dim != 1024) return RL_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(D) & 15) return RL_ERR_UNSUPPORTED;
    const size_t per_query = (size_t)4 * 2 * (dim / 128) * 2 * 64;  // uint4 per query
    const uint4* qfrag = static_cast<const uint4*>(split_buf) + (size_t)first * per_query;
    const float* qmeta = reinterpret_cast<const float*>(static_cast<const char*>(split_buf) + (size_t)n_queries * per_query * 16) + (size_t)first * 8;
    const int64_t tiles = (n_rows + TR - 1) / TR;
    const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles))), blk(512);
#ifdef RAGLITE_EXPERIMENTS  // the tile-timeline build of the kernel exists in experiment builds only
    static unsigned long long* trace = [] {
        unsigned long long* p = nullptr;
        if (exp_env("RAGLITE_HIP_TRACE2")) { (void)hipMalloc(&p, 4096); (void)hipMemset(p, 0, 4096); }
        return p;
    }();
    if (trace && dim == 1024 && !f16) {  // diagnostic build: dump the 30th launch's tile timeline to stderr
        static int calls = 0;
        hipLaunchKernelGGL((maxsim_stream2_kernel<256, true>), grid, blk, 0, s, D, n_rows, qfrag, qmeta, (int)nq, row_to_chunk,
                           chunk_offsets, n_chunks, out, out_stride, split_scale, trace);
        if (++calls == 30) {
            unsigned long long h[512];
            (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
            const unsigned long long t0 = h[0];
            fprintf(stderr, "TRACE2 columns: before-vmcnt after-vmcnt after-B1 after-reads/kreduce after-B2 after-ordinal-DMA after-compute after-walk\n");
            for (int t = 0; t < 8; ++t)
                for (int wv = 0; wv < 8; ++wv) {
                    fprintf(stderr, "TRACE2 tile %d wave %d:", t + 100, wv);
                    for (int k = 0; k < 8; ++k) fprintf(stderr, " %8lld", (long long)(h[(t * 8 + wv) * 8 + k] - t0));
                    fprintf(stderr, "\n");
                }
        }
        return RL_OK;
    }
#endif
#define RL_STREAM2(KW) do { if (f16) hipLaunchKernelGGL((maxsim_stream2_kernel<KW, false, true>), grid, blk, 0, s, D, n_rows, qfrag, qmeta, \
                                                        (int)nq, row_to_chunk, chunk_offsets, n_chunks, out, out_stride, 1.f, nullptr); \
                            else hipLaunchKernelGGL((maxsim_stream2_kernel<KW, false, false>), grid, blk, 0, s, D, n_rows, qfrag, qmeta, \
                                                    (int)nq, row_to_chunk, chunk_offsets, n_chunks, out, out_stride, split_scale, nullptr); } while (0)
    switch (dim) {
        case 128: RL_STREAM2(32); break; case 256: RL_STREAM2(64); break; case 384: RL_STREAM2(96); break;
        case 512: RL_STREAM2(128); break; case 768: RL_STREAM2(192); break; default: RL_STREAM2(256); break;
    }
#undef RL_STREAM2
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_maxsim_stream16(const uint16_t* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                           const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                           float* out, int64_t ld, int n_cu, hipStream_t s) {
    return launch_stream_any(reinterpret_cast<const float*>(D), true, n_rows, dim, Q, nq, row_to_chunk, chunk_offsets,
                             n_chunks, mode, out, ld, n_cu, s);
}

}  // namespace rl
