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
// Design (what the measurements forced -- see DESIGN.md section 5 and profiles/r01_*):
//  * one 512-thread workgroup per CU (LDS-limited), persistent over a contiguous, chunk-aligned row range, so
//    per-chunk maxima never cross workgroups and there is no inter-workgroup hand-off;
//  * WAVE SPECIALISATION.  A wave that issues MFMAs and `global_load_lds` itself stalls on every DMA once the
//    CU's memory queues are full (VMEM issue blocks at memory rate: ~180-370 cycles per 1-KiB DMA measured), and
//    an in-order wave cannot issue MFMAs while it is blocked.  So:
//      waves 0-3  COMPUTE, one per SIMD: K-split (wave w owns columns [256w, 256w+256); its 32 x 256 slice of Q
//                 lives in 128 VGPRs as the MFMA B operand for the whole kernel, so Q costs no LDS traffic).
//                 Per tile: 16 ds_read_b128 (A fragments), 128 MFMAs, one 2-KiB K-partial write.  Nothing else.
//      waves 4-5  LOAD: stream D HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip; one instruction =
//                 one fully coalesced 1-KiB quarter row) into two 16-row stages, always two tiles ahead.  Rows sit
//                 at a 1040-B pitch so the A-operand read (16 lanes = 16 rows, same column) hits 16 different
//                 16-B bank slots.  One loader already saturates a CU's share of HBM; two share the work.
//      waves 6-7  EPILOGUE, alternating tiles: K-reduction of the four partials in a fixed order, segmented
//                 running max over the tile's rows driven by wave-uniform chunk boundaries (hand-issued scalar
//                 loads of row_to_chunk, retired while the wave is parked at a barrier), per-chunk sum over the
//                 query vectors, store.
//  * two workgroup barriers per tile hand the stages back and forth:
//      B1(t): tile t has landed (loaders waited on their own vmcnt) and the K-partials of tile t-1 are in LDS;
//      B2(t): every compute wave holds tile t in registers -> stage t&1 may be refilled with tile t+2.
//    The barriers retire LDS/SMEM operations only (`s_waitcnt lgkmcnt(0); s_barrier`), never VMEM: the DMAs stay
//    in flight across them, counted with `s_waitcnt vmcnt(32)`.
//  * deterministic: fixed K order inside a wave (fp32 MFMA is bitwise an ordered fmaf chain), fixed
//    ((p0+p1)+(p2+p3)) across waves, fixed order of the per-chunk sum.
// Measured on MI355X (32 x 1M x 1024, ragged chunks): 0.805 ms per corpus pass = 5.1 TB/s = 64 % of the 8 TB/s
// peak, matrix pipe ~52 % busy; tile timeline in profiles/r01_tile_timeline.txt.
#include <cstdio>
#include <cstdlib>

#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int TR = 16;      // rows per tile (one 16x16x4 MFMA row block)
constexpr int NSTAGE = 2;
// Compile-time geometry for dim = 4 * KW (KW = columns per compute wave, a multiple of 16).
template <int KW>
struct Geo {
    static constexpr int DIM = 4 * KW;
    static constexpr int QBYTES = KW * 4;                  // one quarter row: KW/4 lanes x 16 B per DMA instruction
    static constexpr int PITCH = QBYTES + 16;              // +16 B: (KW/4 + 1) is odd -> 16 rows hit 16 different bank slots
    static constexpr int STAGE = TR * PITCH;               // per compute wave per stage
    static constexpr int KSTEPS = KW / 16;                 // ds_read_b128 per lane per tile; 4 MFMA k-steps each
    static constexpr int OFF_RED = 4 * NSTAGE * STAGE;
    static constexpr int RED_BYTES = 4 * 2 * 64 * 16;      // 8192 per exchange buffer (double-buffered by tile parity)
    static constexpr int OFF_CM = OFF_RED + 2 * RED_BYTES; // closed-chunk maxima [16][32] fp32
    static constexpr int OFF_STATE = OFF_CM + TR * 32 * 4; // running max of the open chunk [32]
    static constexpr int LDS_TOTAL = OFF_STATE + 32 * 4;   // 151680 B at KW = 256
};

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t target) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// Workgroup barrier that also retires this wave's LDS / SMEM operations, but never its VMEM (the DMAs stay in flight).
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

template <int KW, int NQT, int MODE, bool TRACE = false>
__global__ __launch_bounds__(512, 2) void maxsim_stream_kernel(const float* __restrict__ D, int64_t n_rows,
                                                                 const float* __restrict__ Q, int nq,
                                                                 const int32_t* __restrict__ row_to_chunk,
                                                                 const int64_t* __restrict__ chunk_offsets,
                                                                 int64_t n_chunks, float* __restrict__ out,
                                                                 int64_t ld, unsigned long long* trace) {
    using G_ = Geo<KW>;
    constexpr int SD = G_::DIM, PITCH = G_::PITCH, STAGE = G_::STAGE, KSTEPS = G_::KSTEPS;
    constexpr int OFF_RED = G_::OFF_RED, RED_BYTES = G_::RED_BYTES, OFF_CM = G_::OFF_CM, OFF_STATE = G_::OFF_STATE;
    __shared__ __attribute__((aligned(16))) char smem[G_::LDS_TOTAL];
    const int lane = threadIdx.x & 63;
    const int wv = wave_id();          // 0..7
    // Optional tile timeline (diagnostic build only, RAGLITE_HIP_TRACE=1): workgroup 7, tiles 100..107, 8
    // s_memtime stamps per wave per tile.  Compiled out of the production instantiation.
    auto stamp = [&](int t, int k) {
        if constexpr (TRACE) {
            if (blockIdx.x == 7 && t >= 100 && t < 108 && lane == 0)
                trace[((t - 100) * 8 + wv) * 8 + k] = __builtin_amdgcn_s_memtime();
        }
    };
    const int w = wv & 3;              // K quarter this wave computes (waves 0-3) or feeds (waves 4-7)
    const bool is_loader = wv >= 4;    // wave-uniform
    const int64_t G = gridDim.x, b = blockIdx.x;

    int64_t r_lo, r_hi;
    if constexpr (MODE == 0) {
        const int64_t c_lo = lower_bound_i64(chunk_offsets, n_chunks, (n_rows * b) / G);
        const int64_t c_hi = (b + 1 == G) ? n_chunks : lower_bound_i64(chunk_offsets, n_chunks, (n_rows * (b + 1)) / G);
        r_lo = chunk_offsets[c_lo];
        r_hi = chunk_offsets[c_hi];
    } else {
        const int64_t tiles = (n_rows + TR - 1) / TR;
        r_lo = ((tiles * b) / G) * TR;
        r_hi = ((tiles * (b + 1)) / G) * TR;
        if (r_hi > n_rows) r_hi = n_rows;
    }
    r_lo = uniform_i64(r_lo);
    r_hi = uniform_i64(r_hi);
    const int nt = (int)((r_hi - r_lo + TR - 1) / TR);
    if (nt <= 0) return;  // whole workgroup: no barrier is skipped by a subset of its waves
    const int32_t last_row = (int32_t)(n_rows - 1);  // n_rows < 2^31 (checked by rl_index_create)
    const int32_t r_lo32 = (int32_t)r_lo;
    char* const red_base = smem + OFF_RED;
    constexpr int NQC = 16 * NQT;

    if (!is_loader) {
        // ================================ COMPUTE WAVE ==========================================================
        // Q slice as MFMA B fragments: lane (j = lane & 15, kq = lane >> 4) holds, for MFMA 4*mm + tt,
        // Q[16*h + j][256*w + 16*mm + 4*kq + tt]  (a fixed permutation of K inside the wave).
        const int fj = lane & 15, kq = lane >> 4;
        float qreg[NQT][KW / 4];
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
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
        const char* const a_base = smem + w * NSTAGE * STAGE + fj * PITCH + kq * 16;
        for (int t = 0; t < nt; ++t) {
            stamp(t, 0);
            wg_barrier();  // B1(t): tile t is in stage t&1
            stamp(t, 1);
            f32x4 a[KSTEPS];
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
#pragma unroll
            for (int mm = 0; mm < KSTEPS; ++mm)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int h = 0; h < NQT; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mm][tt], qreg[h][4 * mm + tt], acc[h], 0, 0, 0);
            char* red = red_base + (t & 1) * RED_BYTES;
            stamp(t, 4);
#pragma unroll
            for (int h = 0; h < NQT; ++h) *reinterpret_cast<f32x4*>(red + ((w * 2 + h) * 64 + lane) * 16) = acc[h];
        }
        wg_barrier();  // B1(nt): publishes the K-partials of the last tile
        return;
    }

    if (wv < 6) {
        // ==================================== LOADER WAVE (4, 5) ==================================================
        // One loader already saturates a CU's share of HBM (VMEM issue blocks at memory rate once the CU's request
        // queues are full), so two of the four extra waves stream and never do anything else: loader l feeds the
        // K quarters 2l and 2l+1 (adjacent 1-KiB pieces of each row).
        const int lq = wv - 4;
        const char* const src0 = reinterpret_cast<const char*>(D + KW * (2 * lq));
        const uint32_t lane_off = 16u * lane;
        const bool lane_on = lane < KW / 4;  // a quarter row is KW/4 lanes x 16 B (all 64 lanes at dim 1024)
        auto dma_tile = [&](int t) {  // 2 x 16 quarter rows of tile t -> stage t&1; rows clamped to the corpus
            char* dst0 = smem + ((2 * lq) * NSTAGE + (t & 1)) * STAGE;
            char* dst1 = smem + ((2 * lq + 1) * NSTAGE + (t & 1)) * STAGE;
            const int32_t row0 = r_lo32 + t * TR;
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                int32_t row = row0 + i;
                row = row < last_row ? row : last_row;
                const char* p = src0 + (int64_t)row * (SD * 4) + lane_off;
                if (lane_on) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                     (__attribute__((address_space(3))) void*)(dst0 + i * PITCH), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + KW * 4),
                                                     (__attribute__((address_space(3))) void*)(dst1 + i * PITCH), 16, 0, 0);
                }
            }
        };
        dma_tile(0);
        dma_tile(1);  // rows past the range are clamped: harmless re-reads that keep the vmcnt bookkeeping uniform
        for (int t = 0; t < nt; ++t) {
            stamp(t, 0);
            // Tile t has landed once at most the 32 DMAs of tile t+1 are still outstanding.
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            stamp(t, 1);
            wg_barrier();  // B1(t)
            stamp(t, 2);
            wg_barrier();  // B2(t): stage t&1 is free
            stamp(t, 3);
            dma_tile(t + 2);
            stamp(t, 4);
        }
        wg_barrier();  // B1(nt)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // look-ahead DMAs must not outlive the workgroup's LDS
        return;
    }

    // ==================================== EPILOGUE WAVE (6, 7) ======================================================
    const int ew = wv - 6;
    float* const CM = reinterpret_cast<float*>(smem + OFF_CM);
    float* const STATE = reinterpret_cast<float*>(smem + OFF_STATE);
    const int qc = lane & (NQC - 1);
    const int qh = qc >> 4, qf = qc & 15;

    // MaxSim epilogue of tile te (MODE 0).  C/D layout of 16x16x4: lane (16g + j) of half h holds rows 4g..4g+3 of
    // query 16h + j.  Every lane takes one query column qc and walks the 16 rows with wave-uniform selects.
    // Chunk ordinals of a tile's 16 rows + the row after it (row_to_chunk is padded by 32 entries): scalar loads
    // issued by hand BEFORE the wave parks at B1, retired by that barrier's lgkmcnt(0).
    [[maybe_unused]] i32x8 e_lo, e_hi;
    [[maybe_unused]] int32_t e_last = 0;
    auto load_ordinals = [&](int te) {
        const int32_t* rc = row_to_chunk + (r_lo + (int64_t)te * TR);
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dword %2, %3, 0x40"
                     : "=&s"(e_lo), "=&s"(e_hi), "=&s"(e_last)
                     : "s"(rc)
                     : "memory");
    };
    auto epilogue_maxsim = [&](int te) {
        const int32_t row0 = r_lo32 + te * TR;
        stamp(te + 1, 3);
        int32_t rcv[TR + 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) { rcv[i] = e_lo[i]; rcv[8 + i] = e_hi[i]; }
        rcv[TR] = e_last;
        const int nvalid = ((int32_t)r_hi - row0) < TR ? ((int32_t)r_hi - row0) : TR;
        const char* red = red_base + (te & 1) * RED_BYTES;
        float sv[TR];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx = 16 * g + qf;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + qh) * 64 + idx) * 16);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + ((1 * 2 + qh) * 64 + idx) * 16);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(red + ((2 * 2 + qh) * 64 + idx) * 16);
            const f32x4 p3 = *reinterpret_cast<const f32x4*>(red + ((3 * 2 + qh) * 64 + idx) * 16);
            const f32x4 v = (p0 + p1) + (p2 + p3);  // fixed order: deterministic
            sv[4 * g + 0] = v[0]; sv[4 * g + 1] = v[1]; sv[4 * g + 2] = v[2]; sv[4 * g + 3] = v[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(te + 1, 4);
        float m = (te == 0) ? -INFINITY : STATE[qc];
        int slot = 0;       // chunks closed so far in this tile (wave-uniform)
        int32_t cidv = -1;  // lane s: ordinal of the s-th chunk closed in this tile
        uint32_t ends = 0;  // bit i: row0+i is the last row of its chunk (wave-uniform, scalar registers)
#pragma unroll
        for (int i = 0; i < TR; ++i) ends |= (uint32_t)(rcv[i] != rcv[i + 1]) << i;
        if (nvalid < TR) ends &= (1u << nvalid) - 1u;  // rows past this workgroup's range belong to its neighbour
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            // one v_max per row; the closing work only runs (wave-uniform branch) where a chunk really ends
            m = fmaxf(m, (i < nvalid) ? sv[i] : -INFINITY);
            if ((ends >> i) & 1u) {
                CM[slot * 32 + qc] = m;
                cidv = (lane == slot) ? rcv[i] : cidv;
                m = -INFINITY;
                ++slot;
            }
        }
        STATE[qc] = m;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(te + 1, 5);
        // per-chunk sum over the query vectors: 4 lanes per closed chunk, PER maxima each, then a 2-step butterfly
        const int cs = lane >> 2, part = lane & 3;
        constexpr int PER = NQC / 4;
        const int32_t cid = __shfl(cidv, cs, 64);
        float x = 0.f;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int qq = part * PER + u;
            const float cmv = CM[cs * 32 + qq];
            x += (qq < nq) ? cmv : 0.f;
        }
        stamp(te + 1, 6);
        x += __shfl_xor(x, 1, 64);
        x += __shfl_xor(x, 2, 64);
        if (part == 0 && cs < slot) out[cid] = x;
        stamp(te + 1, 7);
    };
    // Row-score epilogue (MODE 1): finish row group grp (rows 4grp..4grp+3) of tile te for every query column.
    auto epilogue_rows = [&](int te, int grp) {
        if (lane >= NQC) return;
        const int32_t row0 = r_lo32 + te * TR;
        const int nvalid = ((int32_t)r_hi - row0) < TR ? ((int32_t)r_hi - row0) : TR;
        const char* red = red_base + (te & 1) * RED_BYTES;
        const int idx = 16 * grp + qf;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + qh) * 64 + idx) * 16);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + ((1 * 2 + qh) * 64 + idx) * 16);
        const f32x4 p2 = *reinterpret_cast<const f32x4*>(red + ((2 * 2 + qh) * 64 + idx) * 16);
        const f32x4 p3 = *reinterpret_cast<const f32x4*>(red + ((3 * 2 + qh) * 64 + idx) * 16);
        const f32x4 v = (p0 + p1) + (p2 + p3);
        if (qc < nq) {
            float* o = out + (int64_t)qc * ld + row0 + 4 * grp;
            if (4 * grp + 3 < nvalid && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                *reinterpret_cast<f32x4*>(o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * grp + r < nvalid) o[r] = v[r];
            }
        }
    };
    auto epilogue = [&](int te) {
        if constexpr (MODE == 0) {
            if ((te & 1) == ew) epilogue_maxsim(te);  // the two epilogue waves alternate tiles
        } else {
            epilogue_rows(te, 2 * ew);
            epilogue_rows(te, 2 * ew + 1);
        }
    };
    auto barrier_with_ordinals = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(e_lo), "+s"(e_hi), "+s"(e_last)::"memory");
    };
    for (int t = 0; t < nt; ++t) {
        stamp(t, 0);
        const bool mine = MODE == 0 && t > 0 && ((t - 1) & 1) == ew;  // wave-uniform
        if (mine) load_ordinals(t - 1);
        barrier_with_ordinals();  // B1(t): K-partials of tile t-1 are published
        stamp(t, 1);
        wg_barrier();  // B2(t)
        stamp(t, 2);
        if (t > 0) epilogue(t - 1);
    }
    const bool mine_last = MODE == 0 && ((nt - 1) & 1) == ew;
    if (mine_last) load_ordinals(nt - 1);
    barrier_with_ordinals();  // B1(nt)
    epilogue(nt - 1);
}

__global__ __launch_bounds__(256) void row_to_chunk_kernel(const int64_t* __restrict__ chunk_offsets,
                                                            int64_t n_chunks, int64_t n_rows,
                                                            int32_t* __restrict__ row_to_chunk) {
    // One thread per chunk writes its rows' ordinals; rc[n_rows .. n_rows+32] = -1 terminates the last chunk and
    // pads the array so that a tile's 17-entry scalar read never leaves it.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const int64_t b = chunk_offsets[c], e = chunk_offsets[c + 1];
        for (int64_t r = b; r < e; ++r) row_to_chunk[r] = (int32_t)c;
    }
    if (blockIdx.x == 0 && threadIdx.x < 33) row_to_chunk[n_rows + threadIdx.x] = -1;
}

int launch_row_to_chunk(const int64_t* chunk_offsets, int64_t n_chunks, int64_t n_rows, int32_t* row_to_chunk,
                        hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 255) / 256, 4096));
    hipLaunchKernelGGL(row_to_chunk_kernel, dim3(blocks), dim3(256), 0, s, chunk_offsets, n_chunks, n_rows,
                       row_to_chunk);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

namespace {
struct StreamArgs {
    const float* D; int64_t n_rows; const float* Q; int nq; const int32_t* r2c; const int64_t* off; int64_t n_chunks;
    int mode; float* out; int64_t ld; dim3 grid; hipStream_t s; unsigned long long* trace;
};
template <int KW>
void launch_kw(const StreamArgs& a) {
    const dim3 blk(512);
#define RL_STREAM(NQT, MODE) hipLaunchKernelGGL((maxsim_stream_kernel<KW, NQT, MODE>), a.grid, blk, 0, a.s, a.D, a.n_rows, \
                                                a.Q, a.nq, a.r2c, a.off, a.n_chunks, a.out, a.ld, a.trace)
    if (a.mode == 0) { if (a.nq <= 16) RL_STREAM(1, 0); else RL_STREAM(2, 0); }
    else             { if (a.nq <= 16) RL_STREAM(1, 1); else RL_STREAM(2, 1); }
#undef RL_STREAM
}
}  // namespace

// Fast path for dim in {128, 256, 384, 512, 768, 1024} (dim = 4 * KW, KW a multiple of 32) and nq <= 32.
int launch_maxsim_stream(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                         const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                         float* out, int64_t ld, int n_cu, hipStream_t s) {
    if (nq < 1 || nq > 32 || n_rows < 1) return RL_ERR_UNSUPPORTED;
    if (dim != 128 && dim != 256 && dim != 384 && dim != 512 && dim != 768 && dim != 1024) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t tiles = (n_rows + TR - 1) / TR;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles));
    static unsigned long long* trace = [] {
        unsigned long long* p = nullptr;
        if (std::getenv("RAGLITE_HIP_TRACE")) { (void)hipMalloc(&p, 8 * 8 * 8 * 8); (void)hipMemset(p, 0, 8 * 8 * 8 * 8); }
        return p;
    }();
    if (trace && mode == 0 && nq > 16 && dim == 1024) {  // diagnostic build: dump the 30th launch's timeline to stderr
        static int calls = 0;
        hipLaunchKernelGGL((maxsim_stream_kernel<256, 2, 0, true>), dim3(grid), dim3(512), 0, s, D, n_rows, Q, nq,
                           row_to_chunk, chunk_offsets, n_chunks, out, ld, trace);
        if (++calls == 30) {
            unsigned long long h[8 * 8 * 8];
            (void)hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost);
            const unsigned long long t0 = h[0];
            fprintf(stderr, "TRACE columns: compute 0-3: arrive-B1 after-B1 after-A-reads after-B2 after-MFMA | "
                            "loader 4-5: before-vmcnt after-vmcnt after-B1 after-B2 after-DMA-issue | epilogue 6-7: "
                            "arrive-B1 after-B1 after-B2 [epilogue of the previous tile:] start after-K-reduce "
                            "after-row-walk after-chunk-sums done   (shader cycles)\n");
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
    const StreamArgs a{D, n_rows, Q, (int)nq, row_to_chunk, chunk_offsets, n_chunks, mode, out, ld, dim3(grid), s, nullptr};
    switch (dim) {
        case 128: launch_kw<32>(a); break;
        case 256: launch_kw<64>(a); break;
        case 384: launch_kw<96>(a); break;
        case 512: launch_kw<128>(a); break;
        case 768: launch_kw<192>(a); break;
        default: launch_kw<256>(a); break;
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
