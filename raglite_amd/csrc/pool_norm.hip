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
                                                             unsigned int* __restrict__ counter, int tune) {
    constexpr int DIM = 256 * NV, ROWB = DIM * 4, RING = RING_ > 0 ? RING_ : (16 / NV >= 8 ? 8 : 16 / NV), BATCH = 8;
    __shared__ __attribute__((aligned(16))) char smem[WAVES * RING * ROWB];
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

namespace {
// One zeroed 4-byte span counter per launch, from a small per-thread pool of device words (a launch may still be running
// when the same host thread enqueues the next one on another stream).
unsigned int* next_span_counter(hipStream_t s) {
    constexpr int WORDS = 16, SLOTS = 64;  // per launch: 8 span counters (one per XCD)
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
    static const bool vgpr_only = exp_env("RAGLITE_POOL_VGPR") != nullptr;
    if (!vgpr_only && (dim == 256 || dim == 512 || dim == 1024) && n_spans >= 64 && (reinterpret_cast<uintptr_t>(tokens) & 15) == 0 &&
        (!o32 || (reinterpret_cast<uintptr_t>(o32) & 15) == 0) && (!o16 || (reinterpret_cast<uintptr_t>(o16) & 7) == 0)) {
        if (unsigned int* counter = next_span_counter(s)) {
            int n_cu = 256, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
            const int blocks = (int)std::min<int64_t>(n_cu, (n_spans + 63) / 64);
            static const int tune = exp_env("RAGLITE_POOL_BATCH") ? std::atoi(exp_env("RAGLITE_POOL_BATCH")) : 0;
            // (A workgroup-cooperative stream -- one contiguous row range per workgroup, tiles of 8 rows, finisher waves -- streamed at
            // 6.9 TB/s with the spans left unfinished but landed where this kernel is once they were finished, 2.38 vs 2.37 ms, wherever the
            // finishing arithmetic ran: profiles/r02_pool_experiments.txt, DESIGN.md 4.9.  Removed in round 3.)
#define RL_POOL_DMA(NV) \
    hipLaunchKernelGGL((pool_norm_dma_kernel<NV>), dim3(blocks), dim3(512), 0, s, tokens, sb, se, n_spans, normalize, eps, o32, o16, counter, tune)
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
