// Micro-benchmark (round 5): the MAIN LOOP of the MaxSim pass -- corpus slabs through an LDS ring by global_load_lds_dwordx4, query fragments
// straight to registers, fragment reads interleaved with the MFMAs, one barrier per K slab, every wait by count (raglite_amd/csrc/maxsim_pp.hip
// without its tile epilogue) -- for THREE register tiles, to decide whether another tile is worth a rewrite of the kernel:
//
//   current   8 waves (two per SIMD), a wave = 128 rows x 2 queries (64 vectors): 128 accumulator registers;
//             per wave and slab 32 MFMAs, 8 ds_read_b128, 4 global_load_dwordx4, 1 LDS-DMA piece           (24 + 2 operand loads per SIMD and slab)
//   q32       8 waves, a wave = 64 rows x 4 queries: the "32 queries per pass" tile that fits the register file (128 rows x 1024 vectors
//             of accumulators would be the whole file); per wave and slab 32 MFMAs, 4 ds_read_b128, 8 global_load_dwordx4, 1/2 piece
//   w1        4 waves (ONE per SIMD), a wave = 128 rows x 4 queries: 256 accumulator registers (AccVGPRs);
//             per wave and slab 64 MFMAs, 8 ds_read_b128, 8 global_load_dwordx4, 2 pieces                   (16 + 2 per SIMD and slab)
//
// All three stream the same 2-GB image of 1 M x 1024 fp16 rows once per 16 queries ("current", "w1") or once per 32 ("q32") and multiply the
// same number of MFMAs per query; results are not meaningful (no epilogue, nobody reads the sums but one guard store).  Reported: time per
// SIXTEEN queries over the image, to be compared with the kernel's own skeleton (RAGLITE_PP_DBG=128) and with each other on the same box.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tile_shapes.hip -o tile_shapes ; run: ./tile_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int64_t uni64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma(uint32_t lds, const char* src, uint32_t lane16) {
    const uint32_t l = __builtin_amdgcn_readfirstlane(lds);
    const char* const p = reinterpret_cast<const char*>(uni64(reinterpret_cast<int64_t>(src)));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(l), "v"(lane16), "s"(p) : "memory", "m0");
}
template <int OFF>
__device__ __forceinline__ void lds_read(f32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pin(f32x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ h16x8 as_h(const f32x4& v) { h16x8 r; __builtin_memcpy(&r, &v, 16); return r; }
template <int N> using IC = std::integral_constant<int, N>;

constexpr int DC = 8, LC = 6;  // ring depth, look-ahead (slabs)

// WAVES waves per workgroup, QPW queries per wave (2 fragments of 16 vectors each), NBLK 16-row blocks per tile
template <int WAVES, int QPW, int NBLK, bool LAGGED>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void skel(const char* __restrict__ planes, int64_t n_blk, int nslab,
                                                                         const char* __restrict__ qfrag, float* __restrict__ out, int tiles) {
    constexpr int NQF = 2 * QPW;                    // query fragment registers (x 4 VGPRs) per set
    constexpr int CSLOT = NBLK * 1024;              // a corpus slab in LDS
    constexpr int NP = NBLK >= WAVES ? NBLK / WAVES : 1;  // corpus pieces a FEEDING wave fetches per slab
    __shared__ __attribute__((aligned(16))) char smem[DC * CSLOT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool feeds = NBLK >= WAVES || wv < NBLK;  // (wave-uniform)
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;
    const int total = tiles * nslab;
    const int64_t blk0 = (int64_t)blockIdx.x * tiles * NBLK;
    const int64_t slab_bytes = (int64_t)nslab * 1024;
    const char* const qp = qfrag + uni64((int64_t)(wv * QPW) * nslab * 4096);

    int qr_s = 0;
    auto issue_q = [&](f32x4 (&qn)[NQF]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < QPW; ++q) {
            const char* const a = reinterpret_cast<const char*>(uni64(reinterpret_cast<int64_t>(qp + ((int64_t)q * nslab + qr_s) * 4096)));
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:2048"
                         : "=&v"(qn[2 * q]), "=&v"(qn[2 * q + 1]) : "v"(lane16), "s"(a) : "memory");
        }
        if (++qr_s == nslab) qr_s = 0;
    };
    int fc_s = 0, fc_r = 0, fc_slot = 0;
    auto issue_c = [&]() __attribute__((always_inline)) {
        if (feeds) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int piece = NBLK >= WAVES ? wv * NP + p : wv;
                int64_t blk = blk0 + (int64_t)(fc_r < tiles ? fc_r : tiles - 1) * NBLK + piece;
                blk = blk < n_blk ? blk : n_blk - 1;
                dma(lds_base + (uint32_t)(fc_slot * CSLOT + piece * 1024), planes + uni64(blk * slab_bytes + (int64_t)fc_s * 1024), lane16);
            }
        }
        if (++fc_s == nslab) { fc_s = 0; ++fc_r; }
        fc_slot = fc_slot + 1 == DC ? 0 : fc_slot + 1;
    };
    f32x4 acc[NQF][NBLK];
#pragma unroll
    for (int c = 0; c < NQF; ++c)
#pragma unroll
        for (int a = 0; a < NBLK; ++a) acc[c][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ef[NBLK], qA[NQF], qB[NQF];
    int c_slot = 0;
    const uint32_t rd_c = lds_base + lane16;
    auto R = [&](auto A_) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
        lds_read<a * 1024>(ef[a], rd_c + (uint32_t)(c_slot * CSLOT));
    };
    auto G = [&](f32x4 (&q)[NQF], auto A_) __attribute__((always_inline)) {
        constexpr int a = decltype(A_)::value;
#pragma unroll
        for (int c = 0; c < NQF; ++c) acc[c][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(ef[a]), as_h(q[c]), acc[c][a], 0, 0, 0);
    };
    constexpr int H = NBLK / 2;
    auto slab = [&](f32x4 (&q)[NQF], f32x4 (&qn)[NQF], auto LAG_) __attribute__((always_inline)) {
        constexpr bool LAG = decltype(LAG_)::value;
        issue_q(qn);
        issue_c();
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... A>(std::integer_sequence<int, A...>) {
            (([&] {
                 G(q, IC<A>{});
                 if constexpr (!LAG) R(IC<A>{});
                 else if constexpr (A > 0) R(IC<A - 1>{});
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, H>{});
        if constexpr (!LAG) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(H) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(H - 1) : "memory");
        [&]<int... A>(std::integer_sequence<int, A...>) { (pin(ef[H + A]), ...); }(std::make_integer_sequence<int, H>{});
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... A>(std::integer_sequence<int, A...>) {
            (([&] {
                 G(q, IC<H + A>{});
                 if constexpr (!LAG) R(IC<H + A>{});
                 else R(IC<H + A - 1>{});
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, H>{});
        if constexpr (LAG) { R(IC<NBLK - 1>{}); __builtin_amdgcn_sched_barrier(0); }
        c_slot = c_slot + 1 == DC ? 0 : c_slot + 1;
    };
    auto landed = [&](f32x4 (&qn)[NQF]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(H) : "memory");
        if (feeds) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");  // the query fragments have landed when only the pieces behind them are out
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        [&]<int... A>(std::integer_sequence<int, A...>) { (pin(ef[A]), ...); }(std::make_integer_sequence<int, H>{});
        [&]<int... C>(std::integer_sequence<int, C...>) { (pin(qn[C]), ...); }(std::make_integer_sequence<int, NQF>{});
    };
    for (int i = 0; i < LC; ++i) issue_c();
    issue_q(qA);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    [&]<int... A>(std::integer_sequence<int, A...>) { (R(IC<A>{}), ...); }(std::make_integer_sequence<int, NBLK>{});
    c_slot = 1;
    landed(qA);
    auto main_loop = [&](auto LAG_) __attribute__((always_inline)) {
        for (int g = 0; g < total; g += 2) {
            slab(qA, qB, LAG_);
            landed(qB);
            slab(qB, qA, LAG_);
            landed(qA);
        }
    };
    if (!LAGGED || wv < WAVES / 2) main_loop(std::false_type{});
    else main_loop(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NQF; ++c)
#pragma unroll
        for (int a = 0; a < NBLK; ++a) t += acc[c][a];
    if (tiles > 1000000) out[threadIdx.x] = (t[0] + t[1]) + (t[2] + t[3]);
}

// "current" on v_mfma_f32_32x32x16_f16: the same tile (8 waves x 128 rows x 2 queries), the same loads from the same image (the lane -> 16-byte
// granule mapping of a fragment read changes, not the layout: a 32-row x 16-k operand is two half-pieces of the 16-row x 32-k blocks), HALF as
// many MFMA instructions of twice the length: 16 per wave and slab instead of 32, each with 32 cycles of matrix pipe behind it -- an operand
// load issued between two of them has a whole MFMA to hide under even when the SIMD partner has none ready.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool LAGGED>
__global__ __launch_bounds__(512, 2) void skel32(const char* __restrict__ planes, int64_t n_blk, int nslab, const char* __restrict__ qfrag,
                                                  float* __restrict__ out, int tiles) {
    constexpr int NBLK = 8, CSLOT = NBLK * 1024;
    __shared__ __attribute__((aligned(16))) char smem[DC * CSLOT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
    const uint32_t lane16 = 16u * lane;
    // fragment (rb, ks): lane (kg = lane >> 5, r = lane & 31) reads the granule of piece 2 rb + (r >> 4) at index (2 ks + kg) * 16 + (r & 15)
    const uint32_t frag_off = (uint32_t)(((lane & 31) >> 4) * 1024 + (((lane >> 5) * 16 + (lane & 15)) * 16));
    const uint32_t q_off = (uint32_t)(((lane & 31) >> 4) * 2048 + (((lane >> 5) * 16 + (lane & 15)) * 16));
    const int total = tiles * nslab;
    const int64_t blk0 = (int64_t)blockIdx.x * tiles * NBLK;
    const int64_t slab_bytes = (int64_t)nslab * 1024;
    const char* const qp = qfrag + uni64((int64_t)(wv * 2) * nslab * 4096);
    int qr_s = 0;
    auto issue_q = [&](f32x4 (&qn)[4]) __attribute__((always_inline)) {  // [2 * query + ks]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char* const a = reinterpret_cast<const char*>(uni64(reinterpret_cast<int64_t>(qp + ((int64_t)q * nslab + qr_s) * 4096)));
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:512"
                         : "=&v"(qn[2 * q]), "=&v"(qn[2 * q + 1]) : "v"(q_off), "s"(a) : "memory");
        }
        if (++qr_s == nslab) qr_s = 0;
    };
    int fc_s = 0, fc_r = 0, fc_slot = 0;
    auto issue_c = [&]() __attribute__((always_inline)) {
        int64_t blk = blk0 + (int64_t)(fc_r < tiles ? fc_r : tiles - 1) * NBLK + wv;
        blk = blk < n_blk ? blk : n_blk - 1;
        dma(lds_base + (uint32_t)(fc_slot * CSLOT + wv * 1024), planes + uni64(blk * slab_bytes + (int64_t)fc_s * 1024), lane16);
        if (++fc_s == nslab) { fc_s = 0; ++fc_r; }
        fc_slot = fc_slot + 1 == DC ? 0 : fc_slot + 1;
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][rb][i] = 0.f;
    f32x4 ef[8], qA[4], qB[4];  // ef[2 * ks + ... ]: index F = 4 * ks + rb
    int c_slot = 0;
    const uint32_t rd_c = lds_base + frag_off;
    auto R = [&](auto F_) __attribute__((always_inline)) {
        constexpr int F = decltype(F_)::value, ks = F >> 2, rb = F & 3;
        lds_read<rb * 2048 + ks * 512>(ef[F], rd_c + (uint32_t)(c_slot * CSLOT));
    };
    auto G = [&](f32x4 (&q)[4], auto F_) __attribute__((always_inline)) {
        constexpr int F = decltype(F_)::value, ks = F >> 2, rb = F & 3;
        acc[0][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ef[F]), as_h(q[ks]), acc[0][rb], 0, 0, 0);
        acc[1][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ef[F]), as_h(q[2 + ks]), acc[1][rb], 0, 0, 0);
    };
    auto slab = [&](f32x4 (&q)[4], f32x4 (&qn)[4], auto LAG_) __attribute__((always_inline)) {
        constexpr bool LAG = decltype(LAG_)::value;
        issue_q(qn);
        issue_c();
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... A>(std::integer_sequence<int, A...>) {
            (([&] {
                 G(q, IC<A>{});
                 if constexpr (!LAG) R(IC<A>{});
                 else if constexpr (A > 0) R(IC<A - 1>{});
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, 4>{});
        if constexpr (!LAG) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        pin(ef[4]); pin(ef[5]); pin(ef[6]); pin(ef[7]);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... A>(std::integer_sequence<int, A...>) {
            (([&] {
                 G(q, IC<4 + A>{});
                 if constexpr (!LAG) R(IC<4 + A>{});
                 else R(IC<4 + A - 1>{});
                 __builtin_amdgcn_sched_barrier(0);
             }()),
             ...);
        }(std::make_integer_sequence<int, 4>{});
        if constexpr (LAG) { R(IC<7>{}); __builtin_amdgcn_sched_barrier(0); }
        c_slot = c_slot + 1 == DC ? 0 : c_slot + 1;
    };
    auto landed = [&](f32x4 (&qn)[4]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        pin(ef[0]); pin(ef[1]); pin(ef[2]); pin(ef[3]);
        pin(qn[0]); pin(qn[1]); pin(qn[2]); pin(qn[3]);
    };
    for (int i = 0; i < LC; ++i) issue_c();
    issue_q(qA);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    [&]<int... A>(std::integer_sequence<int, A...>) { (R(IC<A>{}), ...); }(std::make_integer_sequence<int, 8>{});
    c_slot = 1;
    landed(qA);
    auto main_loop = [&](auto LAG_) __attribute__((always_inline)) {
        for (int g = 0; g < total; g += 2) {
            slab(qA, qB, LAG_);
            landed(qB);
            slab(qB, qA, LAG_);
            landed(qA);
        }
    };
    if (!LAGGED || wv < 4) main_loop(std::false_type{});
    else main_loop(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) t += acc[q][rb][i];
    if (tiles > 1000000) out[threadIdx.x] = t;
}

__global__ void fill(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u ^ seed;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        p[i] = (h & 0x83FF83FFu) | (((h >> 10) & 3u) + 11u) << 10 | (((h >> 26) & 3u) + 11u) << 26;  // two fp16 values in [-1, 1)
    }
}

template <int WAVES, int QPW, int NBLK, bool LAGGED>
void run(const char* name, const char* planes, int64_t n_rows, int nslab, const char* q, float* out, int queries_per_pass) {
    const int64_t n_blk = n_rows / 16;
    const int64_t n_tiles = n_blk / NBLK;
    const int tiles = (int)((n_tiles + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 8 * 16 / queries_per_pass;  // (a 128-query step)
    for (int i = 0; i < 2; ++i) skel<WAVES, QPW, NBLK, LAGGED><<<dim3(256), dim3(WAVES * 64)>>>(planes, n_blk, nslab, q, out, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < passes * 3; ++i) skel<WAVES, QPW, NBLK, LAGGED><<<dim3(256), dim3(WAVES * 64)>>>(planes, n_blk, nslab, q, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= (float)(passes * 3);
    const double flop = 2.0 * queries_per_pass * 32 * (double)tiles * 256 * NBLK * 16 * (nslab * 32.0);
    printf("%-58s %7.4f ms per pass of %2d queries = %7.4f ms per 16 queries   %7.1f TF   (%s)\n", name, ms, queries_per_pass,
           ms * 16 / queries_per_pass, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

template <bool LAGGED>
void run32(const char* name, const char* planes, int64_t n_rows, int nslab, const char* q, float* out) {
    const int64_t n_blk = n_rows / 16;
    const int tiles = (int)((n_blk / 8 + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) skel32<LAGGED><<<dim3(256), dim3(512)>>>(planes, n_blk, nslab, q, out, tiles);
    hipEventRecord(e0);
    for (int i = 0; i < 24; ++i) skel32<LAGGED><<<dim3(256), dim3(512)>>>(planes, n_blk, nslab, q, out, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 24.f;
    const double flop = 2.0 * 16 * 32 * (double)tiles * 256 * 128 * (nslab * 32.0);
    printf("%-58s %7.4f ms per pass of 16 queries = %7.4f ms per 16 queries   %7.1f TF   (%s)\n", name, ms, ms, flop / ms / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    const int64_t n_rows = 1000000 / 128 * 128 + 128;
    const int nslab = 32;
    char* planes; char* q; float* out;
    hipMalloc(&planes, (size_t)n_rows * 2048 + (1 << 20));
    hipMalloc(&q, (size_t)32 * nslab * 4096);
    hipMalloc(&out, 4096);
    fill<<<2048, 256>>>(reinterpret_cast<uint32_t*>(planes), n_rows * 512, 1u);
    fill<<<256, 256>>>(reinterpret_cast<uint32_t*>(q), (int64_t)32 * nslab * 1024, 2u);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 2, 8, true>("current: 8 waves x (128 rows x 2 queries), lagged", planes, n_rows, nslab, q, out, 16);
        run<8, 2, 8, false>("current without the half-step lag", planes, n_rows, nslab, q, out, 16);
        run<8, 4, 4, true>("q32: 8 waves x (64 rows x 4 queries), lagged", planes, n_rows, nslab, q, out, 32);
        run<4, 4, 8, false>("w1: 4 waves (one per SIMD) x (128 rows x 4 queries)", planes, n_rows, nslab, q, out, 16);
        run32<true>("current tile on v_mfma_f32_32x32x16_f16, lagged", planes, n_rows, nslab, q, out);
        run32<false>("current tile on v_mfma_f32_32x32x16_f16, no lag", planes, n_rows, nslab, q, out);
    }
    return 0;
}
