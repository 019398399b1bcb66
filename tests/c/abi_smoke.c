/* A pure-C consumer of libraglite_hip.so: no Python, no torch, no HIP headers -- only include/raglite_hip.h.
 * Built and run by tests/test_gpu_parity.py::test_pure_c_consumer (gcc; needs an MI355X at run time).
 * Checks, with host pointers: exact cosine top-k on a one-hot corpus (known answer), the two-stage chunk search,
 * a filtered search, append + delete, and MaxSim on the fast path; with device pointers: the synthetic generator and a
 * search that leaves its results on the device. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raglite_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int st_ = (call);                                                                    \
        if (st_ != RL_OK) {                                                                  \
            fprintf(stderr, "%s failed (%d): %s\n", #call, st_, rl_last_error());            \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            fprintf(stderr, "expectation failed: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    enum { N = 640, D = 128, K = 5 };
    int n_dev = 0;
    CHECK(rl_device_count(&n_dev));
    EXPECT(n_dev >= 1);
    CHECK(rl_init(0));
    char name[128];
    int cus = 0;
    int64_t mem = 0;
    CHECK(rl_device_info(0, name, (int)sizeof(name), &cus, &mem));
    printf("device: %s, %d CUs, %.0f GB\n", name, cus, (double)mem / 1e9);

    /* corpus: row r = e_(r % D) scaled by (1 + r / D): cosine ignores the scale, dot does not */
    float* E = (float*)calloc((size_t)N * D, sizeof(float));
    for (int r = 0; r < N; ++r) E[(size_t)r * D + r % D] = 1.0f + (float)(r / D);
    int64_t off[N / 4 + 1]; /* 4 rows per chunk */
    for (int c = 0; c <= N / 4; ++c) off[c] = 4 * c;
    rl_index* idx = NULL;
    CHECK(rl_index_create(&idx, E, N, D, off, N / 4, RL_COSINE, RL_MEM_HOST, NULL));
    int64_t n_rows = 0, n_chunks = 0;
    int32_t dim = 0;
    int metric = -1;
    CHECK(rl_index_info(idx, &n_rows, &dim, &n_chunks, &metric));
    EXPECT(n_rows == N && dim == D && n_chunks == N / 4 && metric == RL_COSINE);

    float q[D];
    memset(q, 0, sizeof(q));
    q[7] = 2.0f; /* cosine 1 with rows 7, 135, 263, 391, 519; 0 with every other row */
    float s[K];
    int32_t r[K];
    CHECK(rl_search_rows(idx, q, 1, K, s, r, RL_MEM_HOST, NULL));
    for (int i = 0; i < K; ++i) {
        EXPECT(r[i] == 7 + 128 * i); /* ties resolve to the lowest row */
        EXPECT(fabsf(s[i] - 1.0f) < 1e-6f);
    }
    float cs[3];
    int32_t cc[3], cnt = 0;
    CHECK(rl_search_chunks(idx, q, 1, 10, 3, cs, cc, &cnt, RL_MEM_HOST, NULL));
    EXPECT(cnt == 3 && cc[0] == 7 / 4 && cc[1] == 135 / 4 && cc[2] == 263 / 4);

    /* filter away chunk 1 (rows 4..7): the best hit moves to row 135 */
    uint32_t filter[(N / 4 + 31) / 32];
    memset(filter, 0xff, sizeof(filter));
    filter[0] &= ~(1u << 1);
    CHECK(rl_search_rows_filtered(idx, q, 1, K, filter, s, r, RL_MEM_HOST, NULL));
    EXPECT(r[0] == 135 && r[3] == 519 && r[4] != 7);

    /* lifecycle: append one chunk holding the query itself, then delete chunk 33 (rows 132..135) */
    CHECK(rl_index_append(idx, q, 1, NULL, 1, RL_MEM_HOST, NULL));
    int64_t dead = 135 / 4;
    CHECK(rl_index_delete_chunks(idx, &dead, 1, NULL));
    int64_t live_rows = 0, live_chunks = 0;
    CHECK(rl_index_live(idx, &live_rows, &live_chunks, NULL));
    EXPECT(live_rows == N + 1 - 4 && live_chunks == N / 4 + 1 - 1);
    CHECK(rl_search_rows(idx, q, 1, K, s, r, RL_MEM_HOST, NULL));
    EXPECT(r[0] == 7 && r[1] == 263 && r[4] == N); /* 135 is gone, the appended row N ties with the others */

    /* arithmetic of the MFMA streaming kernels: this corpus (row maxima 1..5, all finite) gets the fp16 split by default */
    int arith = -1;
    CHECK(rl_index_arithmetic(idx, &arith));
    EXPECT(arith == RL_ARITH_F16_SPLIT);
    CHECK(rl_index_set_arithmetic(idx, RL_ARITH_FP32_EXACT));
    CHECK(rl_index_arithmetic(idx, &arith));
    EXPECT(arith == RL_ARITH_FP32_EXACT);
    EXPECT(rl_index_set_arithmetic(idx, 77) != RL_OK);
    CHECK(rl_index_set_arithmetic(idx, RL_ARITH_AUTO));

    /* MaxSim: two query vectors e_7 and e_9 -> chunk 1 (rows 4..7) scores 1*1 + 0, chunk 2 (rows 8..11) 0 + 1 ... */
    float Q2[2 * D];
    memset(Q2, 0, sizeof(Q2));
    Q2[7] = 1.0f;
    Q2[D + 9] = 1.0f;
    float ms[2];
    int32_t mc[2];
    CHECK(rl_maxsim_topk(idx, Q2, 2, 2, ms, mc, RL_MEM_HOST, NULL));
    /* raw dots: row 519 = 5 * e_7 (chunk 129), row 521 = 5 * e_9 (chunk 130): both score 5 */
    EXPECT(mc[0] == 519 / 4 && mc[1] == 521 / 4 && ms[0] == 5.0f && ms[1] == 5.0f);
    CHECK(rl_index_destroy(idx));

    /* device pointers: generator + search with results left on the device */
    float *dE = NULL, *dq = NULL, *ds = NULL;
    int32_t* dr = NULL;
    CHECK(rl_dev_alloc((void**)&dE, (size_t)4096 * 1024 * sizeof(float)));
    CHECK(rl_dev_alloc((void**)&dq, 1024 * sizeof(float)));
    CHECK(rl_dev_alloc((void**)&ds, K * sizeof(float)));
    CHECK(rl_dev_alloc((void**)&dr, K * sizeof(int32_t)));
    CHECK(rl_synth_fill(dE, 0, (int64_t)4096 * 1024, 42, RL_SYNTH_UNIFORM, NULL));
    CHECK(rl_memcpy_d2d(dq, dE + (size_t)1234 * 1024, 1024 * sizeof(float), NULL)); /* the query IS row 1234 */
    CHECK(rl_index_create(&idx, dE, 4096, 1024, NULL, 0, RL_COSINE, RL_MEM_DEVICE, NULL));
    CHECK(rl_search_rows(idx, dq, 1, K, ds, dr, RL_MEM_DEVICE, NULL));
    CHECK(rl_memcpy_d2h(r, dr, K * sizeof(int32_t), NULL));
    CHECK(rl_memcpy_d2h(s, ds, K * sizeof(float), NULL));
    CHECK(rl_stream_sync(NULL));
    EXPECT(r[0] == 1234 && fabsf(s[0] - 1.0f) < 1e-5f && s[1] < 0.5f);
    CHECK(rl_index_destroy(idx));
    CHECK(rl_dev_free(dE));
    CHECK(rl_dev_free(dq));
    CHECK(rl_dev_free(ds));
    CHECK(rl_dev_free(dr));
    free(E);
    printf("abi_smoke OK (libraglite_hip version %d)\n", rl_version());
    return 0;
}
