"""The pivot route of the half-bytes row search (round 6; hi_filter.hip `transform_bmax_kernel` / `pivot_collect_kernel`, option `hi_pivot`).

`ORDER BY dist LIMIT k` for one to sixteen queries (`/root/reference/src/raglite/_search.py:69-79`): the approximate similarities of the
ranking pass are no longer RANKED -- the candidates are re-scored and ranked exactly anyway -- their threshold comes from the k-th largest of
the workgroup maxima, a lower bound of the k-th best.  Contract: the same rows and the same score bits as the route that ranks first
(`hi_pivot = 0`) and as the full-precision pass (`hi_search = 0`); a candidate list that is a superset of the ranked route's; the guarded
fallback where the bound is defeated.  The route applies for k <= 512 and >= 3 k groups of scores (2048 per workgroup, 512 or 256 per wave)."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, sim_fp32_exact

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("n,dim,B,k", [(70_000, 1024, 1, 10), (140_001, 512, 4, 20), (700_000, 128, 2, 100), (300_003, 256, 16, 48),
                                        (266_240, 256, 1, 1), (800_000, 128, 3, 128),
                                        # k beyond 170: a maximum per WAVE of up to 512 workgroups (G <= 2048) -- the reference's own num_hits are 160 - 256
                                        (400_000, 256, 1, 160), (1_000_000, 128, 2, 256), (600_000, 128, 1, 512), (300_000, 256, 3, 200)])
def test_pivot_route_equals_the_ranked_route_and_the_full_pass(metric, n, dim, B, k):
    E = oracle.synth_matrix(9800 + dim, n, dim)
    Q = oracle.synth_matrix(9810 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    q = Q if B > 1 else Q[0]
    S, R = idx.search_rows(q, k)
    st = idx.filter_stats()
    with idx.options(hi_pivot=0):
        S1, R1 = idx.search_rows(q, k)
        st1 = idx.filter_stats()
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(q, k)
    assert st["kind"] == st1["kind"] == "rows_hi" and not st["fallback"] and not st1["fallback"]
    assert np.array_equal(R, R1) and _same(S, S1)
    assert np.array_equal(R, R0) and _same(S, S0)
    # a superset of the ranked route's candidates (threshold from a lower bound of the k-th best), and not a much larger one
    assert st1["candidates_per_query_max"] >= k
    assert st["candidates_per_query_mean"] >= st1["candidates_per_query_mean"]
    assert st["candidates_per_query_max"] < 1024
    assert st["candidates_per_query_mean"] <= 2.0 * st1["candidates_per_query_mean"] + 64
    S, R = np.atleast_2d(S), np.atleast_2d(R)
    for b in (0, B - 1):
        sims = oracle.similarity(E, Q[b], metric)
        assert_topk_close(S[b], R[b], sims, k, 2e-6 * max(1.0, float(np.abs(sims).max())))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_pivot_route_integer_data_bit_exact_and_ties(metric):
    """Integer-valued data: scores and rows bit-identical to the oracle; thousands of tied similarities around the k-th best (ties -> lowest row)."""
    n, dim, k = 150_000, 256, 24
    E = oracle.synth_matrix(9820, n, dim, "small_int")
    Q = oracle.synth_matrix(9821, 3, dim, "small_int")
    E[5000:9000] = E[4999]  # 4 001 identical rows
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    with idx.options(hi_pivot=0):
        S1, R1 = idx.search_rows(Q, k)
    assert np.array_equal(R, R1) and _same(S, S1)
    for b in range(3):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei)
        assert _same(S[b], es.astype(np.float32))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_pivot_route_defeated_bound_falls_back(metric):
    """3 000 copies of the best row per query: every list overflows, the guarded full-precision pass answers -- the same bits as hi_search = 0."""
    rng = np.random.default_rng(11)
    n, dim, k = 140_000, 512, 16
    E = oracle.synth_matrix(9830, n, dim, "small_int")
    Q = oracle.synth_matrix(9831, 2, dim, "small_int")
    hot = rng.choice(n, 3000, replace=False)
    E[hot] = (2.0 * np.sign(Q.sum(axis=0)))[None, :]
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["fallback"]
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q, k)
    assert np.array_equal(R, R0) and _same(S, S0)
    ref_s, ref_r = oracle.search_rows(E, Q[0], k, metric, np.float64)
    assert np.array_equal(R[0], ref_r)
    idx.close()


def test_pivot_route_where_it_does_not_apply_and_after_append():
    """Too few group maxima (k = 200 over 70 000 rows): the ranked route, silently; rows appended later are searched through the same route."""
    n, dim = 100_000, 256
    E = oracle.synth_matrix(9840, n, dim)
    q = oracle.synth_matrix(9841, 1, dim)[0]
    idx = raglite_amd.DeviceIndex(E[:70_000], metric="cosine")
    for k in (8, 200):
        S, R = idx.search_rows(q, k)
        with idx.options(hi_search=0):
            S0, R0 = idx.search_rows(q, k)
        assert np.array_equal(R, R0) and _same(S, S0)
    idx.append(E[70_000:])
    S, R = idx.search_rows(q, 8)
    ref = raglite_amd.DeviceIndex(E, metric="cosine")
    with ref.options(hi_search=0):
        S0, R0 = ref.search_rows(q, 8)
    assert np.array_equal(R, R0) and _same(S, S0)
    idx.close()
    ref.close()


# ---- the same idea for ONE or TWO MaxSim queries (api.hip: maxsim_few_hi_plane): chunk scores instead of row similarities, a maximum per wave ----
@pytest.mark.parametrize("n_queries,nq,k", [(1, 32, 100), (2, 32, 128), (2, 5, 64), (1, 1, 7)])
def test_few_maxsim_queries_pivot_route_integer_bit_exact(n_queries, nq, k):
    """`sum_i max_j q_i . d_j` per chunk (`/root/reference/src/raglite/_search.py:143-149,394-396` generalised), 100 k chunks of 1-2 rows: the
    route applies (>= 3 k wave maxima); scores and chunk ordinals bit-identical to the oracle and to the ranked route."""
    from tests.util import ragged_offsets

    rng = np.random.default_rng(40 + nq)
    n, dim = 160_000, 512
    off = ragged_offsets(rng, n, 1, 2)
    E = oracle.synth_matrix(9850 + nq, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(9860 + i + nq, nq, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    assert idx.n_chunks >= 3 * k * 256
    S, Cc = idx.maxsim_topk_batch(Qb, k)
    st = idx.filter_stats()
    with idx.options(hi_pivot=0):
        S1, C1 = idx.maxsim_topk_batch(Qb, k)
        st1 = idx.filter_stats()
    assert st["kind"] == st1["kind"] == "maxsim_batch_hi" and not st["fallback"] and not st1["fallback"]
    assert np.array_equal(Cc, C1) and _same(S, S1)
    assert st["candidates_per_query_mean"] >= st1["candidates_per_query_mean"] >= k
    assert st["candidates_per_query_max"] < 2048
    for i in range(n_queries):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(Cc[i], wc) and np.array_equal(S[i], ws)
    s1, c1 = idx.maxsim_topk(Qb[0], k)  # the single-query entry point
    assert np.array_equal(c1, Cc[0]) and _same(s1, S[0])
    idx.close()


def test_few_maxsim_queries_pivot_route_float_data_and_fallback():
    from tests.util import ragged_offsets

    rng = np.random.default_rng(50)
    n, dim, k = 140_000, 512, 100  # (>= 64 M elements: the index keeps a HI plane)
    off = np.arange(n + 1, dtype=np.int64)
    E = oracle.synth_matrix(9870, n, dim)
    Q = oracle.synth_matrix(9871, 8, dim)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s, c = idx.maxsim_topk(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"] and k <= st["candidates_per_query_max"] < 2048
    with idx.options(hi_few=0):
        fs, fc = idx.maxsim_topk(Q, k)
    ref = oracle.maxsim_scores(E, off, Q, np.float64)
    tol = 2e-6 * float(np.abs(ref).max())
    assert_topk_close(s, c, ref, k, tol)
    assert set(c.tolist()) == set(fc.tolist())
    # 4 000 near-identical chunks at the top: the lists overflow, the guarded pass over the rows answers -- the bits of hi_few = 0
    hot = rng.choice(n, 4000, replace=False)
    E2 = E.copy()
    E2[hot] = (3.0 * Q.sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, dim))).astype(np.float32)
    idx2 = raglite_amd.DeviceIndex(E2, off, metric="dot")
    s, c = idx2.maxsim_topk(Q, k)
    st = idx2.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and st["fallback"]
    with idx2.options(hi_few=0):
        fs, fc = idx2.maxsim_topk(Q, k)
    assert np.array_equal(c, fc) and _same(s, fs)
    idx.close()
    idx2.close()


# ---- crowded scores: l2 (select.hip: launch_topk_pivot) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,B,k", [(300_000, 64, 1, 100), (120_000, 128, 3, 10), (400_000, 32, 4, 128), (90_000, 64, 2, 1), (500_000, 32, 2, 256),
                                        (450_000, 64, 1, 512)])
def test_l2_selection_over_crowded_scores_equals_the_radix_selection(n, dim, B, k):
    """`ORDER BY dist LIMIT k` with the l2 metric (`/root/reference/src/raglite/_typing.py:123-134`, `_config.py:69`): the similarities
    1 - |e - q| of a big corpus share their exponent and leading mantissa bits -- one bin of the radix selection holds them all and its exact slow
    path takes milliseconds.  The pivot route (group maxima -> threshold -> a few hundred rows -> ranking) must return the same rows and score bits
    (hi_pivot = 0: the radix selection), the oracle's ranking, and ties to the lowest row."""
    E = oracle.synth_matrix(9900 + dim, n, dim)
    Q = oracle.synth_matrix(9910 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric="l2")
    q = Q if B > 1 else Q[0]
    S, R = idx.search_rows(q, k)
    with idx.options(hi_pivot=0):
        S0, R0 = idx.search_rows(q, k)
    assert np.array_equal(R, R0) and _same(S, S0)
    S, R = np.atleast_2d(S), np.atleast_2d(R)
    for b in (0, B - 1):
        sims = oracle.similarity(E, Q[b], "l2")
        assert_topk_close(S[b], R[b], sims, k, 2e-6 * max(1.0, float(np.abs(sims).max())))
    idx.close()
    # integer data with thousands of exact ties around the k-th distance, and more than 4096 rows at the best distance (the list overflows:
    # the radix selection answers behind the flag)
    Ei = oracle.synth_matrix(9920, n, dim, "small_int")
    Qi = oracle.synth_matrix(9921, 2, dim, "small_int")
    Ei[1000:7000] = Qi[0]
    idx = raglite_amd.DeviceIndex(Ei, metric="l2")
    S, R = idx.search_rows(Qi, k)
    with idx.options(hi_pivot=0):
        S0, R0 = idx.search_rows(Qi, k)
    assert np.array_equal(R, R0) and _same(S, S0)
    assert R[0].tolist() == list(range(1000, 1000 + k))  # distance 0, ties to the lowest row
    for b in range(2):
        es, ei = oracle.topk_desc(sim_fp32_exact(Ei, Qi[b], "l2"), k)
        assert np.array_equal(R[b], ei) and _same(S[b], es.astype(np.float32))
    idx.close()
