"""The pivot route of the half-bytes row search (round 6; hi_filter.hip `transform_bmax_kernel` / `pivot_collect_kernel`, option `hi_pivot`).

`ORDER BY dist LIMIT k` for one to sixteen queries (`/root/reference/src/raglite/_search.py:69-79`): the approximate similarities of the
ranking pass are no longer RANKED -- the candidates are re-scored and ranked exactly anyway -- their threshold comes from the k-th largest of
the workgroup maxima, a lower bound of the k-th best.  Contract: the same rows and the same score bits as the route that ranks first
(`hi_pivot = 0`) and as the full-precision pass (`hi_search = 0`); a candidate list that is a superset of the ranked route's; the guarded
fallback where the bound is defeated.  The route applies for k <= 128 and n >= 3 k x 2048 rows."""

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
                                        (266_240, 256, 1, 1), (800_000, 128, 3, 128)])
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
    """k > 128 or too few workgroup maxima: the ranked route, silently; rows appended later are searched through the same route."""
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
