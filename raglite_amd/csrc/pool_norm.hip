// a1 + a2 + a3: late-chunking mean-pool over token-row spans, L2 normalise, fp16 cast.
//
// Replaces src/raglite/_embed.py:131-140 (and the whole-string variant :154,158-164).  HBM-bound:
// algorithmic bytes = 4*T*dim read + (4 and/or 2)*S*dim written.  One wave per span; a token row of
// dim = 1024 is 4 KiB = four coalesced 1-KiB `global_load_dwordx4` per wave.  Sums are kept in fp64
// (the reference pools float64 arrays) -- at ~10 B/clk/CU of HBM the fp64 adds are <5 % of VALU time.
// mean = sum * (1 / n) and out = mean * (1 / norm): one reciprocal per span instead of a 13-instruction IEEE fp64 divide per
// element -- the 2 x 10^8 divides of a cfg 4 launch were 0.3 ms of its 2.3 (profiles/r02_pool_experiments.txt).  Against
// sum / n / norm that is <= 2 ulp of fp64 (2e-16) more, on top of the summation-order difference to np.mean that was there
// anyway: the fp16 result equals the reference's except where the value sits within ~1e-15 (relative) of an fp16 rounding
// boundary -- 1 element in 10^12 (the golden fixtures, tests/golden/*, stay bit-identical).
#include <cstdlib>

#include "common.h"

namespace rl {

template <int VEC>
struct VecLoad;
template <>
struct VecLoad<4> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        // token rows are read exactly once: non-temporal (no L1 allocation)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
};
template <>
struct VecLoad<1> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = __builtin_nontemporal_load(p); }
};

// fp64 -> fp16 with ONE rounding (RNE), as numpy's astype(float16) does (_embed.py:140).  There is no such instruction and the
// compiler's lowering of (_Float16)double is ~40 integer instructions per element -- 10^8 of them per cfg 4 launch, more
// VALU work than the sums.  Instead: fp64 -> fp32 rounded TO ODD (v_cvt_f32_f64 rounds to nearest; when that was inexact,
// of the two floats around x take the one with the odd significand), then v_cvt_f16_f32: with 13 spare significand bits the
// second rounding of a round-to-odd intermediate equals the direct rounding (Boldo & Melquiond 2008) -- bit-identical for
// every input (checked against numpy on all fp16 midpoints and their fp64 neighbours), NaN stays NaN.
__device__ __forceinline__ uint16_t f64_to_f16_bits(double x) {
    const float f = (float)x;
    const double back = (double)f;  // exact
    uint32_t u = __float_as_uint(f);
    if (back != x && x == x) {      // inexact, not NaN
        const uint32_t other = fabs(back) < fabs(x) ? u + 1u : u - 1u;  // the float on the other side of x
        u = (u & 1u) ? u : other;
    }
    const _Float16 h = (_Float16)__uint_as_float(u);
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}

template <int NV, int VEC>
__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ tokens, int dim,
                                                         const int64_t* __restrict__ span_begin,
                                                         const int64_t* __restrict__ span_end, int64_t n_spans,
                                                         int normalize, double eps, float* __restrict__ out_f32,
                                                         uint16_t* __restrict__ out_f16) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    int col[NV];
    bool ok[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        col[v] = (v * 64 + lane) * VEC;
        ok[v] = col[v] < dim;
    }
    for (int64_t s = wave0; s < n_spans; s += n_waves) {
        const int64_t b = span_begin[s], e = span_end[s];
        double acc[NV][VEC];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[v][j] = 0.0;
        int64_t r = b;
        for (; r + 4 <= e; r += 4) {  // four rows (4*NV loads) in flight per lane
            float x0[NV][VEC], x1[NV][VEC], x2[NV][VEC], x3[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) {
                    VecLoad<VEC>::load(p0 + col[v], x0[v]);
                    VecLoad<VEC>::load(p0 + dim + col[v], x1[v]);
                    VecLoad<VEC>::load(p0 + 2 * (int64_t)dim + col[v], x2[v]);
                    VecLoad<VEC>::load(p0 + 3 * (int64_t)dim + col[v], x3[v]);
                }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {  // same row order as the two-row loop: bit-identical sums
                        acc[v][j] += (double)x0[v][j]; acc[v][j] += (double)x1[v][j];
                        acc[v][j] += (double)x2[v][j]; acc[v][j] += (double)x3[v][j];
                    }
        }
        for (; r + 2 <= e; r += 2) {  // two rows (2*NV loads) in flight per lane
            float x0[NV][VEC], x1[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
            const float* p1 = p0 + dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) { VecLoad<VEC>::load(p0 + col[v], x0[v]); VecLoad<VEC>::load(p1 + col[v], x1[v]); }
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { acc[v][j] += (double)x0[v][j]; acc[v][j] += (double)x1[v][j]; }
        }
        if (r < e) {
            float x0[NV][VEC];
            const float* p0 = tokens + r * (int64_t)dim;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v]) VecLoad<VEC>::load(p0 + col[v], x0[v]);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (ok[v])
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[v][j] += (double)x0[v][j];
        }
        const double rn = 1.0 / (double)(e - b);  // zero rows: 0 * inf = NaN like np.mean of zero rows
        double ss = 0.0;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                acc[v][j] = acc[v][j] * rn;
                if (ok[v]) ss += acc[v][j] * acc[v][j];
            }
        if (normalize) {
            double norm = sqrt(wave_sum(ss));
            if (eps > 0.0) norm = fmax(norm, eps);
            const double rnorm = 1.0 / norm;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[v][j] = acc[v][j] * rnorm;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (!ok[v]) continue;
            if (out_f32) {
                float* o = out_f32 + s * (int64_t)dim + col[v];
                if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(o) =
                        make_float4((float)acc[v][0], (float)acc[v][1], (float)acc[v][2], (float)acc[v][3]);
                } else {
                    o[0] = (float)acc[v][0];
                }
            }
            if (out_f16) {
                uint16_t* o = out_f16 + s * (int64_t)dim + col[v];
                if constexpr (VEC == 4) {
                    ushort4 h;
                    h.x = f64_to_f16_bits(acc[v][0]); h.y = f64_to_f16_bits(acc[v][1]);
                    h.z = f64_to_f16_bits(acc[v][2]); h.w = f64_to_f16_bits(acc[v][3]);
                    *reinterpret_cast<ushort4*>(o) = h;
                } else {
                    o[0] = f64_to_f16_bits(acc[v][0]);
                }
            }
        }
    }
}

template <int NV, int VEC>
static int launch_t(const float* tokens, int32_t dim, const int64_t* sb, const int64_t* se, int64_t n_spans,
                    int normalize, double eps, float* o32, uint16_t* o16, hipStream_t s) {
    // Spans are ragged (4..60 token rows at cfg 4), so a few generations of workgroups balance better than one
    // persistent generation: 8x measured 2.34 ms vs 2.44 ms on 3.2 M token rows.
    const int blocks = (int)std::min<int64_t>((int64_t)persistent_grid(pool_norm_kernel<NV, VEC>, 256, (n_spans + 3) / 4) * 8,
                                              (n_spans + 3) / 4);
    hipLaunchKernelGGL((pool_norm_kernel<NV, VEC>), dim3(blocks), dim3(256), 0, s, tokens, (int)dim, sb, se, n_spans,
                       normalize, eps, o32, o16);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

// Decision shared by the two DMA kernels (see pool_norm_coop_kernel): ordered spans that cover >= 3/4 of their row range.
__device__ __forceinline__ bool pool_layout_cooperative(const unsigned int* __restrict__ layout, int64_t rows_all) {
    const unsigned int bad = __builtin_amdgcn_readfirstlane(layout[0]);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(layout[2]), hi = __builtin_amdgcn_readfirstlane(layout[3]);
    const unsigned long long covered = ((unsigned long long)hi << 32) | lo;
    return bad == 0 && rows_all >= 0 && covered * 4ull >= (unsigned long long)rows_all * 3ull;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant for dim = 256 * NV (bge-m3's 1024 = NV 4): the token rows stream HBM -> LDS by `global_load_lds_dwordx4`
// into a wave-private ring, so the bytes in flight do not live in VGPRs (the register-staged kernel above keeps 64 of its
// 188 VGPRs for them, runs two waves per SIMD and stops streaming at every 4-row group and at every span boundary:
// 5.7 TB/s at BASELINE cfg 4).  Here every wave owns 16 KiB of LDS (ring of 16 / NV rows), refills a slot the moment it
// has read it, and keeps streaming through span boundaries while the previous span is normalised and stored.  No
// workgroup barrier: a wave only ever waits on its own DMA counter.  Spans are claimed from a device counter, eight at a
// time at first and fewer towards the end (ragged spans: a static split leaves a ~20 % tail), results do not depend on who takes which span: rows are
// summed in order in fp64 exactly as above, the mean / norm / cast are the same statements.
// Measured at BASELINE cfg 4 (3.2 M x 1024 rows, same box A/B): 2.24 ms against 2.36 ms for the register-staged kernel.
// What did NOT matter (profiles/r02_pool_experiments.txt): ring depth 3 / 4 / 5 rows per wave, 8 / 4 / 2 waves per CU at equal
// bytes in flight, dropping the fp64 accumulation altogether (same time: the kernel is bound by what the memory system
// delivers to 4-KiB-granular streams, ~6.0 TB/s, not by the waves); dropping `nt` costs 10 %.
template <int NV, int RING_ = 0, bool NT = true, int WAVES = 8>
__global__ __launch_bounds__(64 * WAVES) void pool_norm_dma_kernel(const float* __restrict__ tokens, const int64_t* __restrict__ span_begin,
                                                             const int64_t* __restrict__ span_end, int64_t n_spans, int normalize,
                                                             double eps, float* __restrict__ out_f32, uint16_t* __restrict__ out_f16,
                                                             unsigned int* __restrict__ counter, int tune,
                                                             const unsigned int* __restrict__ layout) {
    constexpr int DIM = 256 * NV, ROWB = DIM * 4, RING = RING_ > 0 ? RING_ : (16 / NV >= 8 ? 8 : 16 / NV), BATCH = 8;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * RING * ROWB];
    if (layout && pool_layout_cooperative(layout, span_end[n_spans - 1] - span_begin[0])) return;  // the cooperative kernel serves
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();
    char* const ring = smem + wv * RING * ROWB;
    const uint32_t ring_lds = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)ring);
    const uint32_t lane16 = 16u * lane;
    const char* const src0 = reinterpret_cast<const char*>(tokens);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int n_queues = gridDim.x < 8 ? (int)gridDim.x : 8;  // one span queue per XCD (fewer on a tiny grid)
    const int xcd = blockIdx.x % n_queues;
    const int64_t x_first = (n_spans * xcd) / n_queues, x_count = (n_spans * (xcd + 1)) / n_queues - x_first;  // this queue's spans
    const int x_waves = (((int)gridDim.x - xcd + n_queues - 1) / n_queues) * WAVES;                               // waves pulling from it
    int64_t x_left = x_count;
    for (;;) {
        // ---- claim up to BATCH spans; lane i keeps span i's bounds ---------------------------------------------------------
        // Guided claiming from one counter per XCD (blockIdx % 8 is the XCD; a single word saturates at a few dozen
        // dequeues per microsecond with 2048 waves pulling): every XCD owns an eighth of the spans; a wave takes BATCH
        // spans while plenty are left and fewer towards the end (a batch of 8 spans is ~250 rows = 15 % of a wave's
        // share: the last batches decide the tail).
        unsigned int base = 0, want = BATCH;
        if (lane == 0) {
            const int64_t share = x_left / (int64_t)(x_waves);
            want = (unsigned int)(share < 1 ? 1 : (share > BATCH ? BATCH : share));
            if (tune > 0) want = (unsigned int)tune;  // experiment: fixed batch size
            base = atomicAdd(counter + xcd, want);
        }
        base = __builtin_amdgcn_readfirstlane(base);
        want = __builtin_amdgcn_readfirstlane(want);
        if ((int64_t)base >= x_count) break;
        x_left = x_count - (int64_t)base - want;  // what was left after this claim (as far as this wave knows)
        const int count = (int)std::min<int64_t>(want, x_count - base);
        base += (unsigned int)x_first;
        int64_t vb = 0, ve = 0;
        if (lane < count) { vb = span_begin[base + lane]; ve = span_end[base + lane]; }
        auto bound = [&](int64_t v, int i) -> int64_t {
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)v, i), hi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)v >> 32), i);
            return (int64_t)(((uint64_t)hi << 32) | lo);
        };
        // ---- fetch cursor: the next row to bring in (skipping empty spans) --------------------------------------------------
        int fs = 0;
        int64_t fr = bound(vb, 0), fe = bound(ve, 0);
        int fetched = 0, consumed = 0;
        auto fetch_skip = [&]() {
            while (fr >= fe && fs + 1 < count) { ++fs; fr = bound(vb, fs); fe = bound(ve, fs); }
        };
        fetch_skip();
        auto fetch_row = [&]() {  // one row (NV pieces of 1 KiB) into slot fetched % RING, if any is left in the batch
            if (fr >= fe) return;
            const char* src = src0 + fr * (int64_t)ROWB;
            const uint32_t lds = ring_lds + (uint32_t)((fetched % RING) * ROWB);
#pragma unroll
            for (int v = 0; v < NV; ++v)
                // (the instruction's immediate offset is added to BOTH the global and the LDS address)
                if constexpr (NT)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 nt" ::"s"(lds), "v"(lane16), "s"(src),
                                 "n"(v * 1024)
                                 : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds), "v"(lane16), "s"(src),
                                 "n"(v * 1024)
                                 : "memory", "m0");
            ++fetched;
            ++fr;
            fetch_skip();
        };
        for (int i = 0; i < RING; ++i) fetch_row();
        // ---- consume span after span ----------------------------------------------------------------------------------------------
        for (int cs = 0; cs < count; ++cs) {
            const int64_t b = bound(vb, cs), e = bound(ve, cs), sidx = (int64_t)base + cs;
            double acc[NV][4];
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[v][j] = 0.0;
            for (int64_t r = b; r < e; ++r) {
                // VMEM retires in order: row `consumed` has landed when at most the rows fetched after it are outstanding
                if (fetched - consumed - 1 == RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * NV) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const char* slot = ring + (consumed % RING) * ROWB;
                f4 x[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) x[v] = *reinterpret_cast<const f4*>(slot + v * 1024 + 16 * lane);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]) : : "memory");  // the slot is read: it may be refilled
                ++consumed;
                fetch_row();
                if (tune == -1) {  // experiment: the memory side alone (one add per piece keeps the reads alive)
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v][0] += (double)x[v][0];
                    continue;
                }
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[v][j] += (double)x[v][j];
            }
            // ---- mean, L2 norm, cast, store: the statements of pool_norm_kernel (same bits) -----------------------------------------
            const double rn = 1.0 / (double)(e - b);  // zero rows: 0 * inf = NaN like np.mean of zero rows
            double ss = 0.0;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[v][j] = acc[v][j] * rn;
                    ss += acc[v][j] * acc[v][j];
                }
            if (normalize) {
                double norm = sqrt(wave_sum(ss));
                if (eps > 0.0) norm = fmax(norm, eps);
                const double rnorm = 1.0 / norm;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[v][j] = acc[v][j] * rnorm;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int col = (v * 64 + lane) * 4;
                if (out_f32)
                    *reinterpret_cast<float4*>(out_f32 + sidx * (int64_t)DIM + col) =
                        make_float4((float)acc[v][0], (float)acc[v][1], (float)acc[v][2], (float)acc[v][3]);
                if (out_f16) {
                    ushort4 h;
                    h.x = f64_to_f16_bits(acc[v][0]); h.y = f64_to_f16_bits(acc[v][1]);
                    h.z = f64_to_f16_bits(acc[v][2]); h.w = f64_to_f16_bits(acc[v][3]);
                    *reinterpret_cast<ushort4*>(out_f16 + sidx * (int64_t)DIM + col) = h;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this batch is in flight when the next one starts
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Workgroup-cooperative variant for ORDERED, (nearly) gap-free spans -- what late chunking produces (_embed.py:119-135:
// consecutive sentences of a segment) and what one-span-per-string pooling is (:154).  The wave-private streams above stay at
// ~6.0 TB/s whatever the ring depth, waves per CU or arithmetic (profiles/r02_pool_experiments.txt): 2 048 independent 4-KiB
// granular streams.  Here a workgroup streams ONE contiguous row range, as the MaxSim stream kernel does at 7 TB/s:
//   * LOADER waves 0-7: tiles of 8 rows (wave w brings row w), a ring of 4 tiles in LDS, one barrier per tile ("epoch");
//     every loader sums a fixed 128-column slice of every row (fp64, rows in order -> the same sums as the kernels above) and,
//     where a span ends, drops its two sums per lane into LDS -- nothing else: measured, every piece of arithmetic on the
//     loaders' path between two barriers costs its full latency (the DMA issue alone stalls a wave ~0.5 k cycles per tile),
//     finishing the spans there made 2.4 ms of 1.9;
//   * FINISHER waves 8-11: finisher (k mod 4) takes span k -- all 1024 sums from LDS after the next barrier -- and runs the
//     statements of pool_norm_dma_kernel on them with that kernel's lane <-> column mapping, alone and without further
//     synchronisation, while the loaders stream on.  The same bits, including the order of the norm reduction.  (Earlier
//     arrangements -- all waves finishing their own columns in one go, a three-stage pipeline over the tile barriers, finishers
//     that each finish a column slice of EVERY span -- all cost the finishing arithmetic in full: profiles/r02_pool_experiments.txt.)
// Two span ends in one epoch insert an extra barrier.  Workgroup ranges are cut at span starts by rows (binary search in
// span_begin), so the work is balanced to within one span without a queue; span bounds are staged through LDS, CAP at a time
// (read from global memory where a span ends they are dependent scalar loads through a memory system busy streaming).
// `layout` = {order violations, -, covered rows (u64)} from pool_layout_kernel: the kernel returns at once unless the spans are
// ordered and cover >= 3/4 of the row range they span (the wave-private kernel then runs instead, guarded the other way).
template <int NV>
__global__ __launch_bounds__(768) void pool_norm_coop_kernel(const float* __restrict__ tokens, const int64_t* __restrict__ span_begin,
                                                               const int64_t* __restrict__ span_end, int64_t n_spans, int normalize,
                                                               double eps, float* __restrict__ out_f32, uint16_t* __restrict__ out_f16,
                                                               const unsigned int* __restrict__ layout, int dbg) {
    // dbg (RAGLITE_POOL_DBG, timing experiments only): 1 = no LDS reads / adds, 2 = spans are not finished, 4 = the finishers
    // skip their work, 32 = the loaders do not deposit
    constexpr int DIM = 256 * NV, ROWB = DIM * 4, TILE = 8, RING = 4, CAP = 256, LOADERS = 8, FINISHERS = 4;
    constexpr int OFF_DEP = RING * TILE * ROWB;       // [2][DIM] fp64: the loaders' sums of a finished span
    constexpr int OFF_TAB = OFF_DEP + 2 * DIM * 8;    // [2][CAP] int64: span bounds
    __shared__ __attribute__((aligned(16))) char smem[OFF_TAB + 2 * CAP * 8];
    auto uni = [](int64_t v) -> int64_t {
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const int64_t R0 = uni(span_begin[0]), rows_all = uni(span_end[n_spans - 1]) - R0;
    if (!pool_layout_cooperative(layout, rows_all)) return;  // whole grid
    const int lane = threadIdx.x & 63, wv = wave_id();
    const bool loader = wv < LOADERS;  // wave-uniform
    const int cw = wv & (LOADERS - 1);  // the column slice this wave sums (loader) or finishes (finisher)
    if (loader && !(dbg & 128)) __builtin_amdgcn_s_setprio(3);  // the stream comes first wherever a loader and a finisher compete for issue
    const int64_t G = gridDim.x, b = blockIdx.x;
    auto first_span_at = [&](int64_t row) -> int64_t {  // first span with begin >= row (spans are ordered)
        int64_t lo = 0, hi = n_spans;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (uni(span_begin[mid]) >= row) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const int64_t s_lo = b == 0 ? 0 : first_span_at(R0 + (rows_all * b) / G);
    const int64_t s_hi = b + 1 == G ? n_spans : first_span_at(R0 + (rows_all * (b + 1)) / G);
    if (s_lo >= s_hi) return;  // whole workgroup
    const int64_t r_begin = uni(span_begin[s_lo]), r_end = uni(span_end[s_hi - 1]);
    const int64_t n_tiles = (r_end - r_begin + TILE - 1) / TILE;
    const uint32_t ring_lds = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;
    double* const dep = reinterpret_cast<double*>(smem + OFF_DEP);
    int64_t* const tb = reinterpret_cast<int64_t*>(smem + OFF_TAB);
    int64_t* const te = tb + CAP;
    // ---- the stream: loader w brings row w of every tile; exactly NV DMAs per loader and tile (rows past the range re-read the
    // last row into a free slot) so that the in-order vmcnt bookkeeping is the same for every tile --------------------------------
    auto fetch_tile = [&](int64_t t) __attribute__((always_inline)) {
        int64_t row = r_begin + t * TILE + cw;
        row = row < r_end ? row : r_end - 1;
        const char* src = reinterpret_cast<const char*>(tokens) + uni(row) * (int64_t)ROWB;
        const uint32_t lds = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)((((int)(t % RING)) * TILE + cw) * ROWB));
#pragma unroll
        for (int v = 0; v < NV; ++v)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3 nt" ::"s"(lds), "v"(lane16), "s"(src), "n"(v * 1024)
                         : "memory", "m0");
    };
    // ---- this lane's two columns: slice cw = the columns that lanes 8 cw .. 8 cw + 7 of pool_norm_kernel own ---------------------
    const int vq = lane >> 4, li = lane & 15;
    const bool active = vq < NV;
    const int c0 = (active ? vq : 0) * 256 + 32 * cw + 2 * li;
    double acc0 = 0.0, acc1 = 0.0;
    // ---- span bounds, CAP at a time in LDS ----------------------------------------------------------------------------------------
    int64_t s = s_lo, s_base = s_lo;
    auto load_table = [&]() __attribute__((always_inline)) {  // spans s_base .. s_base + CAP - 1 (workgroup-uniform call)
        __syncthreads();  // nobody still reads the previous batch
        for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
            const int64_t j = s_base + i;
            tb[i] = j < s_hi ? span_begin[j] : 0;
            te[i] = j < s_hi ? span_end[j] : 0;
        }
        __syncthreads();
    };
    load_table();
    constexpr int64_t NO_SPAN = INT64_MAX;
    auto fetch_bounds = [&](int64_t idx, int64_t& b_, int64_t& e_) __attribute__((always_inline)) {
        if (idx >= s_hi) { b_ = NO_SPAN; e_ = NO_SPAN; return; }
        if (idx - s_base >= CAP) { s_base = idx; load_table(); }
        b_ = uni(tb[idx - s_base]);
        e_ = uni(te[idx - s_base]);
    };
    int64_t cb, ce, nb, ne;  // the open span s and the one after it (read a whole span before it is needed)
    fetch_bounds(s, cb, ce);
    fetch_bounds(s + 1, nb, ne);
    // ---- finishing.  A span's sums are dropped into one of two LDS slots by the loaders (p0: dropped this epoch, visible after
    // the next barrier); finisher (k mod 4) then takes span k ALONE, with the lane <-> column mapping and the statements of
    // pool_norm_dma_kernel -- the same bits -- while everybody else streams on: one span in four per finisher. ------------------
    unsigned k0 = 0;  // spans dropped so far
    bool p0 = false;  // (workgroup-uniform)
    int64_t p0_s = 0, p0_n = 0;
    unsigned p0_par = 0, p0_k = 0;
    // The finisher's job, in pieces of a few hundred cycles, ONE piece per epoch: it takes part in every tile barrier, and a
    // job done in one go (~2.3 us) holds the next two barriers -- and with them the loaders' DMA issue -- up by what it
    // exceeds the epoch (1.2 us) by: that is how finishing cost its full price on waves that do nothing else.
    int f_stage = 0;  // 0 = idle (wave-uniform)
    int64_t f_s = 0;
    double f_acc[NV][4], f_ss = 0.0, f_rnorm = 1.0;
    auto finisher_step = [&]() __attribute__((always_inline)) {
        switch (f_stage) {
            case 1: {  // the butterfly of the norm (the xor order of wave_sum: same bits)
                f_ss = wave_sum(f_ss);
                f_stage = 2;
                break;
            }
            case 2: {
                double norm = sqrt(f_ss);
                if (eps > 0.0) norm = fmax(norm, eps);
                f_rnorm = 1.0 / norm;
                f_stage = 3;
                break;
            }
            case 3:
            case 4: {  // scale, cast, store: half of the columns per epoch
                const int v0 = f_stage == 3 ? 0 : (NV + 1) / 2, v1 = f_stage == 3 ? (NV + 1) / 2 : NV;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (v < v0 || v >= v1) continue;
                    const int col = (v * 64 + lane) * 4;
                    double o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = normalize ? f_acc[v][j] * f_rnorm : f_acc[v][j];
                    if (out_f32)
                        *reinterpret_cast<float4*>(out_f32 + f_s * (int64_t)DIM + col) = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
                    if (out_f16) {
                        ushort4 h;
                        h.x = f64_to_f16_bits(o[0]); h.y = f64_to_f16_bits(o[1]);
                        h.z = f64_to_f16_bits(o[2]); h.w = f64_to_f16_bits(o[3]);
                        *reinterpret_cast<ushort4*>(out_f16 + f_s * (int64_t)DIM + col) = h;
                    }
                }
                f_stage = f_stage == 3 && NV > 1 ? 4 : 0;
                break;
            }
            default: break;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {  // right after a workgroup barrier
        if (!loader && !(dbg & 4)) {
            finisher_step();
            if (p0 && wv - LOADERS == (int)(p0_k % FINISHERS)) {
                while (f_stage != 0) finisher_step();  // still busy (spans shorter than a tile, four in a row): finish in one go
                const double* const slot = dep + p0_par * DIM;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const double2 a = *reinterpret_cast<const double2*>(slot + (v * 64 + lane) * 4);
                    const double2 c = *reinterpret_cast<const double2*>(slot + (v * 64 + lane) * 4 + 2);
                    f_acc[v][0] = a.x; f_acc[v][1] = a.y; f_acc[v][2] = c.x; f_acc[v][3] = c.y;
                }
                const double rn = 1.0 / (double)p0_n;  // zero rows: 0 * inf = NaN like np.mean of zero rows
                double ss = 0.0;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f_acc[v][j] = f_acc[v][j] * rn;
                        ss += f_acc[v][j] * f_acc[v][j];
                    }
                f_ss = ss;
                f_s = p0_s;
                f_stage = normalize ? 1 : 3;
            }
        }
        p0 = false;
    };
    auto finish_span = [&]() __attribute__((always_inline)) {  // span s = [cb, ce) is complete (workgroup-uniform)
        if (p0) {  // the previous span ended in this epoch too: an epoch of its own
            __syncthreads();
            advance();
        }
        if (loader && active) {
            double2 sums;
            sums.x = acc0; sums.y = acc1;
            *reinterpret_cast<double2*>(dep + (k0 & 1u) * DIM + c0) = sums;
        }
        acc0 = 0.0;
        acc1 = 0.0;
        p0_s = s; p0_n = ce - cb; p0_par = k0 & 1u; p0_k = k0;
        p0 = true;
        ++k0;
        ++s;
        cb = nb;
        ce = ne;
        fetch_bounds(s + 1, nb, ne);
    };
    if (n_tiles > 0 && loader)  // (a range of empty spans only has no rows to stream)
        for (int t = 0; t < RING - 1; ++t) fetch_tile(t);
    for (int64_t t = 0; t < n_tiles; ++t) {
        // VMEM retires in order: at most the NV DMAs of tiles t + 1 and t + 2 outstanding <=> this loader's row of tile t landed
        if (loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NV) : "memory");
        __syncthreads();  // every row of tile t is in LDS; everybody is done with tile t - 1
        if (loader) fetch_tile(t + RING - 1);  // into the slot of tile t - 1
        advance();
        const char* const tile = smem + ((int)(t % RING)) * TILE * ROWB;
        const int64_t row0 = r_begin + t * TILE;
        const int64_t row1 = row0 + TILE < r_end ? row0 + TILE : r_end;
        if (dbg & 1) continue;
        if (!(dbg & 2))
            while (row0 >= ce) finish_span();  // the open span ended with the previous tile (also: empty spans on the way)
        const bool inside = row0 >= cb && row1 <= ce;
        const bool one_cut = row0 >= cb && ce < row1 && nb == ce && ne >= row1;  // one span end inside, the next span adjoins
        const bool sum = loader && active;
        if (inside || one_cut || (dbg & 2)) {
            // All LDS reads first, then the adds in row order (a read-then-add per row exposes the LDS latency eight times).  With
            // one span end inside the tile: rows before it, the deposit, rows after it -- all under uniform branches (an add of a
            // masked 0.0 would turn a sum of -0.0 into +0.0).
            const int cut = (inside || (dbg & 2)) ? TILE : (int)(ce - row0);  // 1 .. TILE - 1 with one_cut
            const int live = (int)(row1 - row0);                               // TILE except in the range's last tile
            float2 x[TILE];
            if (sum) {
#pragma unroll
                for (int i = 0; i < TILE; ++i) x[i] = *reinterpret_cast<const float2*>(tile + i * ROWB + c0 * 4);
#pragma unroll
                for (int i = 0; i < TILE; ++i)
                    if (i < cut && i < live) {
                        acc0 += (double)x[i].x;
                        acc1 += (double)x[i].y;
                    }
            }
            if (cut < TILE) {
                finish_span();
                if (sum) {
#pragma unroll
                    for (int i = 1; i < TILE; ++i)
                        if (i >= cut && i < live) {
                            acc0 += (double)x[i].x;
                            acc1 += (double)x[i].y;
                        }
                }
            }
        } else {  // several span ends or a gap inside the tile: row by row
            float2 x = make_float2(0.f, 0.f);
            if (sum) x = *reinterpret_cast<const float2*>(tile + c0 * 4);  // one row ahead of the walk
#pragma unroll 1
            for (int i = 0; i < TILE; ++i) {
                const int64_t row = row0 + i;
                if (row >= r_end) break;
                float2 xn = x;
                if (sum) xn = *reinterpret_cast<const float2*>(tile + (i + 1 < TILE ? i + 1 : i) * ROWB + c0 * 4);
                while (row >= ce) finish_span();
                if (row >= cb && sum) {  // (rows in a gap between spans are skipped)
                    acc0 += (double)x.x;
                    acc1 += (double)x.y;
                }
                x = xn;
            }
        }
    }
    while (s < s_hi && !(dbg & 3)) finish_span();  // the last span of the range and empty spans after it
    if (p0) {  // the last span
        __syncthreads();
        advance();
    }
    while (f_stage != 0) finisher_step();  // (per wave: no barrier involved)
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
}

// {order violations, -, covered rows (u64)} of a span list, for the guards of the two DMA kernels.
__global__ __launch_bounds__(256) void pool_layout_kernel(const int64_t* __restrict__ span_begin, const int64_t* __restrict__ span_end,
                                                           int64_t n_spans, unsigned int* __restrict__ layout) {
    unsigned int bad = 0;
    unsigned long long covered = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_spans; i += (int64_t)gridDim.x * 256) {
        const int64_t b = span_begin[i], e = span_end[i];
        bad += (e < b) || (i + 1 < n_spans && span_begin[i + 1] < e);
        covered += (unsigned long long)(e > b ? e - b : 0);
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o, 64); covered += __shfl_xor(covered, o, 64); }
    if ((threadIdx.x & 63) == 0 && (bad || covered)) {
        if (bad) atomicAdd(layout, bad);
        atomicAdd(reinterpret_cast<unsigned long long*>(layout + 2), covered);
    }
}

namespace {
// One zeroed 4-byte span counter per launch, from a small per-thread pool of device words (a launch may still be running
// when the same host thread enqueues the next one on another stream).
unsigned int* next_span_counter(hipStream_t s) {
    constexpr int WORDS = 16, SLOTS = 64;  // per launch: 8 span counters (one per XCD) + 4 words of span-layout summary
    static thread_local unsigned int* pool = nullptr;
    static thread_local int at = 0;
    if (!pool && hipMalloc(&pool, SLOTS * WORDS * sizeof(unsigned int)) != hipSuccess) { pool = nullptr; (void)hipGetLastError(); return nullptr; }
    unsigned int* c = pool + WORDS * (at++ % SLOTS);
    if (hipMemsetAsync(c, 0, WORDS * sizeof(unsigned int), s) != hipSuccess) return nullptr;
    return c;
}
}  // namespace

int launch_pool_norm(const float* tokens, int32_t dim, const int64_t* sb, const int64_t* se, int64_t n_spans,
                     int normalize, double eps, float* o32, uint16_t* o16, hipStream_t s) {
    if (n_spans <= 0) return RL_OK;
    // dim = 256 * NV with fp32-vector-aligned buffers: the LDS-DMA stream (RAGLITE_POOL_VGPR=1 keeps the register-staged kernel)
    static const bool vgpr_only = std::getenv("RAGLITE_POOL_VGPR") != nullptr;
    if (!vgpr_only && (dim == 256 || dim == 512 || dim == 1024) && n_spans >= 64 && (reinterpret_cast<uintptr_t>(tokens) & 15) == 0 &&
        (!o32 || (reinterpret_cast<uintptr_t>(o32) & 15) == 0) && (!o16 || (reinterpret_cast<uintptr_t>(o16) & 7) == 0)) {
        if (unsigned int* counter = next_span_counter(s)) {
            int n_cu = 256, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
            const int blocks = (int)std::min<int64_t>(n_cu, (n_spans + 63) / 64);
            static const int tune = std::getenv("RAGLITE_POOL_BATCH") ? std::atoi(std::getenv("RAGLITE_POOL_BATCH")) : 0;
            // The workgroup-cooperative stream (pool_norm_coop_kernel) is opt-in: RAGLITE_POOL_COOP=1 (read per call: tests flip it).
            // Measured on the cfg 4 shape it streams at 6.9 TB/s with the spans left unfinished but lands where the wave-private
            // kernel is (2.38 vs 2.37 ms, same box) once they are finished -- wherever the finishing arithmetic runs
            // (profiles/r02_pool_experiments.txt) -- so the simpler kernel stays the default.
            const char* co = std::getenv("RAGLITE_POOL_COOP");
            const bool no_coop = !(co && co[0] && co[0] != '0');
            unsigned int* layout = no_coop ? nullptr : counter + 8;
            const char* dbg_env = std::getenv("RAGLITE_POOL_DBG");
            const int dbg = dbg_env ? std::atoi(dbg_env) : 0;
            if (layout)
                hipLaunchKernelGGL(pool_layout_kernel, dim3((unsigned)std::min<int64_t>(64, (n_spans + 1023) / 1024)), dim3(256), 0, s, sb, se, n_spans, layout);
#define RL_POOL_DMA(NV)                                                                                                                     \
    do {                                                                                                                                    \
        if (layout) hipLaunchKernelGGL((pool_norm_coop_kernel<NV>), dim3(blocks), dim3(768), 0, s, tokens, sb, se, n_spans, normalize, eps, o32, o16, layout, dbg); \
        hipLaunchKernelGGL((pool_norm_dma_kernel<NV>), dim3(blocks), dim3(512), 0, s, tokens, sb, se, n_spans, normalize, eps, o32, o16, counter, tune, layout); \
    } while (0)
            if (dim == 256) RL_POOL_DMA(1); else if (dim == 512) RL_POOL_DMA(2); else RL_POOL_DMA(4);
#undef RL_POOL_DMA
            RL_HIP(hipGetLastError());
            return RL_OK;
        }
    }
    const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(tokens) & 15) == 0) &&
                      (!o32 || (reinterpret_cast<uintptr_t>(o32) & 15) == 0) &&
                      (!o16 || (reinterpret_cast<uintptr_t>(o16) & 7) == 0);
#define RL_POOL_CASE(NV, VEC) return launch_t<NV, VEC>(tokens, dim, sb, se, n_spans, normalize, eps, o32, o16, s)
    if (vec4) {
        const int nv = (dim + 255) / 256;
        if (nv <= 1) RL_POOL_CASE(1, 4);
        if (nv <= 2) RL_POOL_CASE(2, 4);
        if (nv <= 3) RL_POOL_CASE(3, 4);
        if (nv <= 4) RL_POOL_CASE(4, 4);
        if (nv <= 6) RL_POOL_CASE(6, 4);
        if (nv <= 8) RL_POOL_CASE(8, 4);
        if (nv <= 16) RL_POOL_CASE(16, 4);
    } else {
        const int nv = (dim + 63) / 64;
        if (nv <= 1) RL_POOL_CASE(1, 1);
        if (nv <= 2) RL_POOL_CASE(2, 1);
        if (nv <= 4) RL_POOL_CASE(4, 1);
        if (nv <= 8) RL_POOL_CASE(8, 1);
        if (nv <= 16) RL_POOL_CASE(16, 1);
        if (nv <= 32) RL_POOL_CASE(32, 1);
    }
#undef RL_POOL_CASE
    return fail(RL_ERR_UNSUPPORTED, "rl_pool_norm: dim must be <= 4096 (multiple of 4) or <= 2048 otherwise");
}

}  // namespace rl
