/* Pure-C probe of the experimental switches (no Python / torch start-up: the whole run is a few seconds of GPU time).
 *   RAGLITE_HI_ONE_PRODUCT  approximate MaxSim pass at one fp16 MFMA product per multiply (read per call)
 *   RAGLITE_HI_RNE          HI halves rounded to nearest (read when an index is created)
 *   RAGLITE_FUSED_HI        big-batch row top-k over the HI image (read per call)
 * For each: results against the shipped default path on the benchmark shape, kernel-only pass times (rl_time_kernel), step times.
 * Build:  gcc -O2 -std=c11 -Iinclude scripts/micro/r3_probe.c -o gpurun_out/r3_probe -Lraglite_amd/_lib -lraglite_hip -lm \
 *             -Wl,-rpath,'$ORIGIN/../raglite_amd/_lib'
 * Run from the repo root on the GPU box:  gpurun_out/r3_probe [rows] */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "raglite_hip.h"

#define CHECK(call)                                                               \
    do {                                                                          \
        int st_ = (call);                                                         \
        if (st_ != RL_OK) {                                                       \
            printf("FAILED %s (%d): %s\n", #call, st_, rl_last_error());          \
            fflush(stdout);                                                       \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

enum { D = 1024, NQ = 32, QB = 128, K = 100, B2 = 1000 };

static int cmp_i32(const void* a, const void* b) { return (*(const int32_t*)a > *(const int32_t*)b) - (*(const int32_t*)a < *(const int32_t*)b); }

/* how two result sets relate: identical bits / same id sets per query / worst relative score difference */
static void compare(const char* what, const float* s0, const int32_t* i0, const float* s1, const int32_t* i1, int nq, int k) {
    int bit_equal = memcmp(s0, s1, (size_t)nq * k * 4) == 0 && memcmp(i0, i1, (size_t)nq * k * 4) == 0;
    int set_diff = 0;
    double worst = 0.0;
    int32_t* a = (int32_t*)malloc((size_t)k * 4);
    int32_t* b = (int32_t*)malloc((size_t)k * 4);
    for (int q = 0; q < nq; ++q) {
        memcpy(a, i0 + (size_t)q * k, (size_t)k * 4);
        memcpy(b, i1 + (size_t)q * k, (size_t)k * 4);
        qsort(a, (size_t)k, 4, cmp_i32);
        qsort(b, (size_t)k, 4, cmp_i32);
        for (int j = 0; j < k; ++j) set_diff += a[j] != b[j];
        for (int j = 0; j < k; ++j) {
            const double d = fabs((double)s0[(size_t)q * k + j] - (double)s1[(size_t)q * k + j]);
            const double m = fabs((double)s0[(size_t)q * k + j]) + 1e-30;
            if (d / m > worst) worst = d / m;
        }
    }
    free(a);
    free(b);
    printf("  %-34s bit-identical %d, sorted-id mismatches %d of %d, worst relative score difference (by rank) %.3g\n", what, bit_equal, set_diff,
           nq * k, worst);
    fflush(stdout);
}

static int maxsim_steps(rl_index* idx, const float* dq, float* ds, int32_t* dc, int steps, double* ms_per_step) {
    CHECK(rl_maxsim_topk_batch(idx, dq, QB, NQ, K, ds, dc, RL_MEM_DEVICE, NULL)); /* warm */
    CHECK(rl_stream_sync(NULL));
    const double t0 = now_ms();
    for (int i = 0; i < steps; ++i) CHECK(rl_maxsim_topk_batch(idx, dq, QB, NQ, K, ds, dc, RL_MEM_DEVICE, NULL));
    CHECK(rl_stream_sync(NULL));
    *ms_per_step = (now_ms() - t0) / steps;
    return 0;
}

static int rows_steps(rl_index* idx, const float* dq, float* ds, int32_t* dr, int steps, double* ms_per_batch) {
    CHECK(rl_search_rows(idx, dq, B2, K, ds, dr, RL_MEM_DEVICE, NULL)); /* warm */
    CHECK(rl_stream_sync(NULL));
    const double t0 = now_ms();
    for (int i = 0; i < steps; ++i) CHECK(rl_search_rows(idx, dq, B2, K, ds, dr, RL_MEM_DEVICE, NULL));
    CHECK(rl_stream_sync(NULL));
    *ms_per_batch = (now_ms() - t0) / steps;
    return 0;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
    const double t_start = now_ms();
    CHECK(rl_init(0));
    /* ragged chunks of 1..15 rows */
    int64_t* off = (int64_t*)malloc((size_t)(N + 2) * sizeof(int64_t));
    int64_t n_chunks = 0, r = 0;
    off[0] = 0;
    while (r < N) {
        int64_t sz = 1 + (int64_t)((n_chunks * 2654435761u) >> 7) % 15;
        if (r + sz > N) sz = N - r;
        r += sz;
        off[++n_chunks] = r;
    }
    void *dE = NULL, *dQ = NULL, *dQ2 = NULL, *dS = NULL, *dC = NULL, *dS2 = NULL, *dR2 = NULL;
    CHECK(rl_dev_alloc(&dE, (size_t)N * D * 4));
    CHECK(rl_dev_alloc(&dQ, (size_t)QB * NQ * D * 4));
    CHECK(rl_dev_alloc(&dQ2, (size_t)B2 * D * 4));
    CHECK(rl_dev_alloc(&dS, (size_t)QB * K * 4));
    CHECK(rl_dev_alloc(&dC, (size_t)QB * K * 4));
    CHECK(rl_dev_alloc(&dS2, (size_t)B2 * K * 4));
    CHECK(rl_dev_alloc(&dR2, (size_t)B2 * K * 4));
    CHECK(rl_synth_fill((float*)dE, 0, N * D, 1234, RL_SYNTH_UNIFORM, NULL));
    CHECK(rl_synth_fill((float*)dQ, 0, (int64_t)QB * NQ * D, 99, RL_SYNTH_UNIFORM, NULL));
    CHECK(rl_synth_fill((float*)dQ2, 0, (int64_t)B2 * D, 77, RL_SYNTH_UNIFORM, NULL));
    float *s0 = malloc((size_t)QB * K * 4), *s1 = malloc((size_t)QB * K * 4);
    int32_t *c0 = malloc((size_t)QB * K * 4), *c1 = malloc((size_t)QB * K * 4);
    float *rs0 = malloc((size_t)B2 * K * 4), *rs1 = malloc((size_t)B2 * K * 4);
    int32_t *rr0 = malloc((size_t)B2 * K * 4), *rr1 = malloc((size_t)B2 * K * 4);
    double ms = 0.0;
    float kms = 0.f;
    const char* debug = getenv("R3_PROBE_DEBUG");

    for (int rne = 0; rne < 2; ++rne) {
        setenv("RAGLITE_HI_RNE", rne ? "1" : "0", 1);
        setenv("RAGLITE_HI_ONE_PRODUCT", "0", 1);
        setenv("RAGLITE_FUSED_HI", "0", 1);
        rl_index* idx = NULL;
        const double tc = now_ms();
        CHECK(rl_index_create(&idx, (const float*)dE, N, D, off, n_chunks, RL_DOT, RL_MEM_DEVICE, NULL));
        CHECK(rl_stream_sync(NULL));
        printf("index rne=%d: %lld rows, %lld chunks, created in %.0f ms (t = %.1f s)\n", rne, (long long)N, (long long)n_chunks, now_ms() - tc,
               (now_ms() - t_start) / 1e3);
        fflush(stdout);
        /* kernel-only pass times */
        for (int kind = 3; kind <= 6; ++kind) {
            if (kind == 4) continue;
            if (rl_time_kernel(idx, kind, (const float*)dQ, 8 * NQ, 2, &kms, NULL) != RL_OK) { printf("  kind %d: %s\n", kind, rl_last_error()); continue; }
            CHECK(rl_time_kernel(idx, kind, (const float*)dQ, 8 * NQ, 10, &kms, NULL));
            printf("  pass kind %d (3: full precision, 5: HI two products, 6: HI one product): %.4f ms\n", kind, kms / 10);
        }
        fflush(stdout);
        if (getenv("R3_PROBE_PASSES_ONLY")) { CHECK(rl_index_destroy(idx)); break; }  /* (timing experiments with RAGLITE_GEMM_DBG) */
        /* MaxSim batch: two products (shipped) vs one product */
        if (maxsim_steps(idx, (const float*)dQ, (float*)dS, (int32_t*)dC, 3, &ms)) return 1;
        CHECK(rl_memcpy_d2h(rne ? s1 : s0, dS, (size_t)QB * K * 4, NULL));
        CHECK(rl_memcpy_d2h(rne ? c1 : c0, dC, (size_t)QB * K * 4, NULL));
        CHECK(rl_stream_sync(NULL));
        printf("  maxsim batch, two products:  %.3f ms per %d-query step = %.0f queries/s\n", ms, QB, QB / ms * 1e3);
        if (rne) compare("rne two products vs rtz default:", s0, c0, s1, c1, QB, K);
        if (debug) { setenv("RAGLITE_HI_DEBUG", "1", 1); CHECK(rl_maxsim_topk_batch(idx, (const float*)dQ, QB, NQ, K, (float*)dS, (int32_t*)dC, RL_MEM_DEVICE, NULL)); unsetenv("RAGLITE_HI_DEBUG"); }
        setenv("RAGLITE_HI_ONE_PRODUCT", "1", 1);
        if (maxsim_steps(idx, (const float*)dQ, (float*)dS, (int32_t*)dC, 3, &ms)) return 1;
        CHECK(rl_memcpy_d2h(s1, dS, (size_t)QB * K * 4, NULL));
        CHECK(rl_memcpy_d2h(c1, dC, (size_t)QB * K * 4, NULL));
        CHECK(rl_stream_sync(NULL));
        printf("  maxsim batch, ONE product:   %.3f ms per %d-query step = %.0f queries/s\n", ms, QB, QB / ms * 1e3);
        compare(rne ? "rne one product vs rtz default:" : "rtz one product vs rtz default:", s0, c0, s1, c1, QB, K);
        if (debug) { setenv("RAGLITE_HI_DEBUG", "1", 1); CHECK(rl_maxsim_topk_batch(idx, (const float*)dQ, QB, NQ, K, (float*)dS, (int32_t*)dC, RL_MEM_DEVICE, NULL)); unsetenv("RAGLITE_HI_DEBUG"); }
        setenv("RAGLITE_HI_ONE_PRODUCT", "0", 1);
        /* big batch of single-vector queries (cfg 5 like, dot metric): shipped fused top-k vs the HI-image variant */
        if (!rne) {
            if (rows_steps(idx, (const float*)dQ2, (float*)dS2, (int32_t*)dR2, 2, &ms)) return 1;
            CHECK(rl_memcpy_d2h(rs0, dS2, (size_t)B2 * K * 4, NULL));
            CHECK(rl_memcpy_d2h(rr0, dR2, (size_t)B2 * K * 4, NULL));
            CHECK(rl_stream_sync(NULL));
            printf("  rows batch (B = %d, k = %d), shipped fused top-k:      %.3f ms\n", B2, K, ms);
        }
        for (int two = 0; two < 2; ++two) {
            setenv("RAGLITE_FUSED_HI", "1", 1);
            setenv("RAGLITE_FUSED_HI_TWO_PRODUCTS", two ? "1" : "0", 1);
            if (rows_steps(idx, (const float*)dQ2, (float*)dS2, (int32_t*)dR2, 2, &ms)) return 1;
            CHECK(rl_memcpy_d2h(rs1, dS2, (size_t)B2 * K * 4, NULL));
            CHECK(rl_memcpy_d2h(rr1, dR2, (size_t)B2 * K * 4, NULL));
            CHECK(rl_stream_sync(NULL));
            printf("  rows batch, HI image, %s:            %.3f ms\n", two ? "two products" : "ONE product ", ms);
            compare("HI-image fused top-k vs shipped:", rs0, rr0, rs1, rr1, B2, K);
        }
        setenv("RAGLITE_FUSED_HI", "0", 1);
        CHECK(rl_index_destroy(idx));
        printf("  (t = %.1f s)\n", (now_ms() - t_start) / 1e3);
        fflush(stdout);
    }
    printf("r3_probe done in %.1f s\n", (now_ms() - t_start) / 1e3);
    return 0;
}
