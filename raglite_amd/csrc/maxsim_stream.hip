// a9 (metric shape) and batched a6: the MFMA streaming tile kernel, dim == 1024.
//
//   mode 0 (MaxSim):  out[c]            = sum_{i<nq} max_{j in chunk c} Q[i].D[j]      (nq <= 32)
//   mode 1 (rows):    out[i*ld + row]   = Q[i].D[row]  (raw dots; scan.hip:transform_kernel applies the
//                                         metric of src/raglite/_typing.py:123-134 afterwards)
// The multi-query generalisation of src/raglite/_search.py:143-149 / src/raglite/_query_adapter.py:174.
//
// Roofline: per corpus pass 4*N*1024 B of HBM traffic against 2*nq*N*1024 flop
// of exact-fp32 MFMA (157.3 TF peak): at nq = 32 the two are within 20 % of each other (0.51 ms vs
// 0.42 ms per 1 M rows), so the kernel must overlap them; at nq <= 16 it is purely HBM-bound.
//
// Design (CDNA4-specific):
//  * one 256-thread workgroup per CU (LDS-limited), persistent over a contiguous, chunk-aligned row
//    range, so per-chunk maxima never cross workgroups and no inter-workgroup hand-off exists;
//  * K-split: wave w owns columns [256w, 256w+256).  Its slice of Q (32 x 256 fp32) lives in 128 VGPRs as
//    the MFMA B operand for the whole kernel -- Q costs no LDS reads and no re-streaming;
//  * D rows are streamed HBM -> LDS with `global_load_lds_dwordx4` (no VGPR round trip): one wave
//    instruction moves one 1-KiB quarter row (fully coalesced); rows sit in LDS at a 1040-B pitch so the
//    per-lane `ds_read_b128` of the A operand (16 lanes = 16 different rows, same column) hits 16
//    distinct 16-B bank slots;
//  * 16-row tiles (v_mfma_f32_16x16x4_f32: 32-cycle issue, two independent accumulators for the two
//    16-query halves), two 16-KiB stages per wave = 128 KiB of the 160-KiB LDS; the DMA of tile t+1/t+2
//    stays in flight across the compute of tile t (counted `s_waitcnt vmcnt(16)`, never 0 in steady
//    state, raw `s_barrier`);
//  * the four K-partials meet in LDS once per tile; one wave (rotating) reduces them and runs the
//    epilogue: a segmented running max over the tile's rows driven by wave-uniform chunk boundaries
//    (scalar loads of row_to_chunk), closed chunks' per-query maxima summed and stored.
//  * fp32 MFMA is bitwise an ordered fmaf chain (cdna_hip_programming.md section 3), so results are
//    deterministic: fixed K order inside a wave, fixed ((p0+p1)+(p2+p3)) across waves.
#include "common.h"

namespace rl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int SD = 1024;                      // embedding dimension of the fast path
constexpr int TR = 16;                        // rows per tile
constexpr int PITCH = 1040;                   // LDS bytes per staged quarter row (1024 + 16 pad)
constexpr int STAGE = TR * PITCH;             // 16640 B per wave per stage
constexpr int NSTAGE = 2;
constexpr int OFF_RED = 4 * NSTAGE * STAGE;   // 133120
constexpr int RED_BYTES = 4 * 2 * 64 * 16;    // one K-partial exchange buffer (4 waves x 2 halves)
constexpr int OFF_S = OFF_RED + 2 * RED_BYTES;   // 149504: S tile [16 rows][33] fp32
constexpr int S_PITCH = 33;
constexpr int OFF_CM = OFF_S + TR * S_PITCH * 4; // 151616: closed-chunk maxima [16][32] fp32
constexpr int OFF_STATE = OFF_CM + TR * 32 * 4;  // 153664: running max of the open chunk [32]
constexpr int OFF_CID = OFF_STATE + 32 * 4;      // 153792: closed-chunk ordinals [16]
constexpr int LDS_TOTAL = OFF_CID + TR * 4;      // 153856 B  (<= 163840)
}  // namespace

// First c in [0, n] with a[c] >= target (a ascending, n+1 entries).
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

template <int NQT, int MODE>
__global__ __launch_bounds__(256, 1) void maxsim_stream_kernel(const float* __restrict__ D, int64_t n_rows,
                                                                const float* __restrict__ Q, int nq,
                                                                const int32_t* __restrict__ row_to_chunk,
                                                                const int64_t* __restrict__ chunk_offsets,
                                                                int64_t n_chunks, float* __restrict__ out,
                                                                int64_t ld) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_TOTAL];
    const int lane = threadIdx.x & 63;
    const int w = wave_id();
    const int64_t G = gridDim.x, b = blockIdx.x;

    // ---- this workgroup's row range -----------------------------------------------------------------
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
    if (nt <= 0) return;

    // ---- Q slice as MFMA B fragments: lane (j = lane & 15, kq = lane >> 4) holds, for MFMA 4*mm + tt,
    //      Q[16*h + j][256*w + 16*mm + 4*kq + tt]  (the K order inside a wave is a fixed permutation) ------
    const int fj = lane & 15, kq = lane >> 4;
    float qreg[NQT][64];
#pragma unroll
    for (int h = 0; h < NQT; ++h) {
        const int qi = 16 * h + fj;
        const int qc_ = qi < nq ? qi : nq - 1;  // clamped load, zeroed below: padded query vectors add 0
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Q + (int64_t)qc_ * SD + 256 * w + 16 * mm + 4 * kq);
            if (qi >= nq) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            qreg[h][4 * mm + 0] = v[0]; qreg[h][4 * mm + 1] = v[1];
            qreg[h][4 * mm + 2] = v[2]; qreg[h][4 * mm + 3] = v[3];
        }
    }

    // ---- HBM -> LDS DMA of one tile's quarter rows (16 x 1 KiB per wave) -------------------------------
    const float* lane_src = D + 256 * w + 4 * lane;
    auto issue = [&](int t, int s) {
        char* dst = smem + (w * NSTAGE + s) * STAGE;
        const int64_t row0 = r_lo + (int64_t)t * TR;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            int64_t row = row0 + i;
            if (row > n_rows - 1) row = n_rows - 1;  // clamp: padded rows are computed and ignored
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(lane_src + row * SD),
                (__attribute__((address_space(3))) void*)(dst + i * PITCH), 16, 0, 0);
        }
    };

    issue(0, 0);
    if (nt > 1) issue(1, 1);

    const char* a_base = smem + w * NSTAGE * STAGE + fj * PITCH + kq * 16;
    float* const S = reinterpret_cast<float*>(smem + OFF_S);
    float* const CM = reinterpret_cast<float*>(smem + OFF_CM);
    float* const STATE = reinterpret_cast<float*>(smem + OFF_STATE);
    int32_t* const CID = reinterpret_cast<int32_t*>(smem + OFF_CID);
    constexpr int NQC = 16 * NQT;  // query columns carried through the epilogue

    // Chunk ordinals of a tile's 16 rows + the row after it (17 dwords) are fetched with hand-issued scalar
    // loads ONE TILE AHEAD: hipcc would otherwise either wait for them in front of the LDS reads (SMEM and LDS
    // share lgkmcnt) or, for a vector load, drain the DMA pipeline with vmcnt(0).  Issued right after the
    // A-fragment wait, their latency hides behind the 128 MFMAs; the barrier's lgkmcnt(0) retires them.
    // row_to_chunk is padded by 32 entries so that the look-ahead never leaves the array.
    [[maybe_unused]] i32x8 rc_lo, rc_hi, rn_lo, rn_hi;
    [[maybe_unused]] int32_t rc_last = 0, rn_last = 0;
    auto load_rc = [&](int64_t row, i32x8& lo, i32x8& hi, int32_t& last) {
        const int32_t* rc = row_to_chunk + row;
        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dword %2, %3, 0x40"
                     : "=&s"(lo), "=&s"(hi), "=&s"(last)
                     : "s"(rc)
                     : "memory");
    };
    if constexpr (MODE == 0) load_rc(r_lo, rc_lo, rc_hi, rc_last);

    for (int t = 0; t < nt; ++t) {
        const int s = t & 1;
        // Tile t has landed once at most the 16 DMAs of tile t+1 are still outstanding.
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        const int64_t row0 = r_lo + (int64_t)t * TR;
        f32x4 acc[NQT];
#pragma unroll
        for (int h = 0; h < NQT; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* ap = a_base + s * STAGE;
        // All 16 A fragments (64 VGPRs) are requested up front so that one wave per SIMD keeps the matrix
        // pipe busy behind counted lgkmcnt waits instead of paying the LDS latency every 16 MFMAs.
        f32x4 a[16];
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) a[mm] = *reinterpret_cast<const f32x4*>(ap + mm * 64);
        if constexpr (MODE == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rc_lo), "+s"(rc_hi), "+s"(rc_last)::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);  // keep hipcc from sinking the reads back next to their MFMAs
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int h = 0; h < NQT; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mm][tt], qreg[h][4 * mm + tt], acc[h], 0, 0, 0);
            if constexpr (MODE == 0) {
                if (mm == 0) {
                    // Next tile's chunk ordinals: issued behind the first MFMAs (i.e. behind hipcc's own wait
                    // for the A fragments), pinned so the wait cannot end up after it.
                    __builtin_amdgcn_sched_barrier(0);
                    load_rc(row0 + TR, rn_lo, rn_hi, rn_last);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // Stage s is consumed: refill it with tile t+2 before anything else.
        if (t + 2 < nt) issue(t + 2, s);

        // K-partials -> LDS, one barrier per tile (exchange buffer double-buffered by tile parity).
        char* red = smem + OFF_RED + (t & 1) * RED_BYTES;
#pragma unroll
        for (int h = 0; h < NQT; ++h) *reinterpret_cast<f32x4*>(red + ((w * 2 + h) * 64 + lane) * 16) = acc[h];
        if constexpr (MODE == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(rn_lo), "+s"(rn_hi), "+s"(rn_last)::"memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // this tile's ordinals for the epilogue; the look-ahead becomes current for the next iteration
        [[maybe_unused]] const i32x8 ec_lo = rc_lo, ec_hi = rc_hi;
        [[maybe_unused]] const int32_t ec_last = rc_last;
        if constexpr (MODE == 0) { rc_lo = rn_lo; rc_hi = rn_hi; rc_last = rn_last; }

        if (w != (t & 3)) continue;  // epilogue duty rotates over the four waves
        // C/D layout of 16x16x4: column (lane & 15) = query, row = 4*(lane >> 4) + reg = corpus row in tile.
        f32x4 v[NQT];
#pragma unroll
        for (int h = 0; h < NQT; ++h) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + h) * 64 + lane) * 16);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + ((1 * 2 + h) * 64 + lane) * 16);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(red + ((2 * 2 + h) * 64 + lane) * 16);
            const f32x4 p3 = *reinterpret_cast<const f32x4*>(red + ((3 * 2 + h) * 64 + lane) * 16);
            v[h] = (p0 + p1) + (p2 + p3);
        }
        const int nvalid = (int)((r_hi - row0) < TR ? (r_hi - row0) : TR);

        if constexpr (MODE == 1) {
#pragma unroll
            for (int h = 0; h < NQT; ++h) {
                const int qi = 16 * h + fj;
                if (qi >= nq) continue;
                float* o = out + (int64_t)qi * ld + row0 + 4 * kq;
                if (4 * kq + 3 < nvalid && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                    *reinterpret_cast<f32x4*>(o) = v[h];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kq + r < nvalid) o[r] = v[h][r];
                }
            }
        } else {
            int32_t rcv[TR + 1];
#pragma unroll
            for (int i = 0; i < 8; ++i) { rcv[i] = ec_lo[i]; rcv[8 + i] = ec_hi[i]; }
            rcv[TR] = ec_last;
            uint32_t ends = 0;  // bit i: row0+i is the last row of its chunk (wave-uniform)
#pragma unroll
            for (int i = 0; i < TR; ++i) ends |= (uint32_t)(rcv[i] != rcv[i + 1]) << i;
#pragma unroll
            for (int h = 0; h < NQT; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) S[(4 * kq + r) * S_PITCH + 16 * h + fj] = v[h][r];
            const int qc = lane & (NQC - 1);
            float m = (t == 0) ? -INFINITY : STATE[qc];
            float sv[TR];
#pragma unroll
            for (int i = 0; i < TR; ++i) sv[i] = S[i * S_PITCH + qc];  // one batch of LDS reads, one wait
            int ncl = 0;
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                if (i < nvalid) {  // wave-uniform
                    m = fmaxf(m, sv[i]);
                    if ((ends >> i) & 1u) {  // row0+i closes chunk rcv[i] (wave-uniform)
                        CM[ncl * 32 + qc] = m;
                        if (lane == 0) CID[ncl] = rcv[i];
                        ++ncl;
                        m = -INFINITY;
                    }
                }
            }
            STATE[qc] = m;
            // Sum the per-query maxima of every chunk closed in this tile: 4 lanes per chunk.
            const int slot = lane >> 2, part = lane & 3;
            constexpr int PER = NQC / 4;
            if (slot < ncl) {
                float x = 0.f;
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int qq = part * PER + u;
                    x += (qq < nq) ? CM[slot * 32 + qq] : 0.f;
                }
                x += __shfl_xor(x, 1, 64);
                x += __shfl_xor(x, 2, 64);
                if (part == 0) out[CID[slot]] = x;
            }
        }
    }
}

__global__ __launch_bounds__(256) void row_to_chunk_kernel(const int64_t* __restrict__ chunk_offsets,
                                                            int64_t n_chunks, int64_t n_rows,
                                                            int32_t* __restrict__ row_to_chunk) {
    // One thread per chunk writes its rows' ordinals; rc[n_rows .. n_rows+32] = -1 terminates the last chunk
    // and pads the array so that a tile's 17-entry scalar read never leaves it.
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += stride) {
        const int64_t b = chunk_offsets[c], e = chunk_offsets[c + 1];
        for (int64_t r = b; r < e; ++r) row_to_chunk[r] = (int32_t)c;
    }
    if (blockIdx.x == 0 && threadIdx.x < 33) row_to_chunk[n_rows + threadIdx.x] = -1;  // terminator + padding
}

int launch_row_to_chunk(const int64_t* chunk_offsets, int64_t n_chunks, int64_t n_rows, int32_t* row_to_chunk,
                        hipStream_t s) {
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 255) / 256, 4096));
    hipLaunchKernelGGL(row_to_chunk_kernel, dim3(blocks), dim3(256), 0, s, chunk_offsets, n_chunks, n_rows,
                       row_to_chunk);
    RL_HIP(hipGetLastError());
    return RL_OK;
}

int launch_maxsim_stream(const float* D, int64_t n_rows, int32_t dim, const float* Q, int32_t nq,
                         const int32_t* row_to_chunk, const int64_t* chunk_offsets, int64_t n_chunks, int mode,
                         float* out, int64_t ld, int n_cu, hipStream_t s) {
    if (dim != SD || nq < 1 || nq > 32 || n_rows < 1) return RL_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(D) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return RL_ERR_UNSUPPORTED;
    const int64_t tiles = (n_rows + TR - 1) / TR;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_cu > 0 ? n_cu : 256, tiles));
    const dim3 g(grid), blk(256);
    if (mode == 0) {
        if (nq <= 16)
            hipLaunchKernelGGL((maxsim_stream_kernel<1, 0>), g, blk, 0, s, D, n_rows, Q, nq, row_to_chunk, chunk_offsets,
                               n_chunks, out, ld);
        else
            hipLaunchKernelGGL((maxsim_stream_kernel<2, 0>), g, blk, 0, s, D, n_rows, Q, nq, row_to_chunk, chunk_offsets,
                               n_chunks, out, ld);
    } else {
        if (nq <= 16)
            hipLaunchKernelGGL((maxsim_stream_kernel<1, 1>), g, blk, 0, s, D, n_rows, Q, nq, row_to_chunk, chunk_offsets,
                               n_chunks, out, ld);
        else
            hipLaunchKernelGGL((maxsim_stream_kernel<2, 1>), g, blk, 0, s, D, n_rows, Q, nq, row_to_chunk, chunk_offsets,
                               n_chunks, out, ld);
    }
    RL_HIP(hipGetLastError());
    return RL_OK;
}

}  // namespace rl
