"""GPU parity of the half-bytes single-query search (api.hip `search_rows_hi`): for up to four queries over a big fp32 corpus
the ranking pass streams only the HI halves of the fp16 split (2 B per element), a rigorous error bound turns its top-2048
into a candidate set that contains the exact top-k, and the candidates are re-scored by the exact kernels.

Contract: the same bits as the full-precision pass (`ORDER BY dist LIMIT k`, `/root/reference/src/raglite/_search.py:69-79`,
ranked exactly) -- checked against the option hi_search = 0 bit for bit, against the oracle, and on corpora built to defeat the
bound (thousands of near-duplicates of the best row), where the guarded full-precision pass has to answer."""

import os

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, sim_fp32_exact

pytestmark = pytest.mark.gpu



def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("n,dim,B,k", [(70_000, 1024, 1, 100), (66_000, 1024, 4, 512), (140_000, 512, 2, 10), (530_000, 128, 3, 100),
                                        (70_000, 1024, 16, 100), (68_000, 1024, 9, 40)])
def test_hi_search_equals_full_pass_bitwise(metric, n, dim, B, k):
    E = oracle.synth_matrix(9500 + dim, n, dim)
    Q = oracle.synth_matrix(9600 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q if B > 1 else Q[0], k)
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q if B > 1 else Q[0], k)
    assert np.array_equal(R, R0) and _same(S, S0)
    S, R = np.atleast_2d(S), np.atleast_2d(R)
    for b in (0, B - 1):
        sims = oracle.similarity(E, Q[b], metric)
        assert_topk_close(S[b], R[b], sims, k, 2e-6 * max(1.0, float(np.abs(sims).max())))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_hi_search_integer_data_bit_exact(metric):
    n, dim, k = 80_000, 1024, 64
    E = oracle.synth_matrix(9700, n, dim, "small_int")
    Q = oracle.synth_matrix(9701, 3, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    for b in range(3):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei)
        assert _same(S[b], es.astype(np.float32))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_near_duplicates_defeat_the_bound_and_the_full_pass_answers(metric):
    """5000 rows within 1e-4 of the query's best match: more than 2048 rows sit inside twice the error bound of the k-th score,
    the guard flags it, the full-precision pass ranks them -- same bits as without the HI plane."""
    rng = np.random.default_rng(11)
    n, dim, k = 70_000, 1024, 100
    E = oracle.synth_matrix(9800, n, dim)
    q = oracle.synth_matrix(9801, 1, dim)[0]
    dup = rng.choice(n, 5000, replace=False)
    E[dup] = (q[None, :] * 0.9 + 1e-4 * rng.standard_normal((5000, dim))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(q, k)
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(q, k)
    assert np.array_equal(R, R0) and _same(S, S0)
    assert np.isin(R, dup).all()
    idx.close()


def test_hi_plane_follows_append_and_two_stage_search():
    n, dim = 70_000, 1024
    E = oracle.synth_matrix(9900, n + 3000, dim)
    q = oracle.synth_matrix(9901, 1, dim)[0]
    idx = raglite_amd.DeviceIndex(E[:n], metric="cosine")
    idx.append(E[n:])
    S, R = idx.search_rows(q, 50)
    ref = raglite_amd.DeviceIndex(E, metric="cosine")
    with ref.options(hi_search=0):
        S0, R0 = ref.search_rows(q, 50)
    assert np.array_equal(R, R0) and _same(S, S0)
    cs, cc, cn = idx.search_chunks(q, 40, 5)  # every row its own chunk here: the two-stage search rides on the same path
    cs0, cc0, cn0 = ref.search_chunks(q, 40, 5)
    assert cn == cn0 and np.array_equal(cc, cc0) and _same(cs, cs0)
    idx.close()
    ref.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_hi_search_with_filter_and_tombstones(metric):
    """Metadata filter / deleted chunks: masked rows rank -inf in the approximate pass, so they are neither in its top-k nor
    among the candidates; a filter that leaves fewer than k rows makes the bound unusable and the full pass answers."""
    rng = np.random.default_rng(21)
    n, dim, k = 70_000, 1024, 100
    E = oracle.synth_matrix(9950, n, dim)
    Q = oracle.synth_matrix(9951, 2, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    ok = rng.random(n) < 0.4
    S, R = idx.search_rows(Q, k, chunk_filter=ok)
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q, k, chunk_filter=ok)
    assert np.array_equal(R, R0) and _same(S, S0) and ok[R].all()
    dead = np.unique(R[:, :30])
    idx.delete_chunks(dead)
    S1, R1 = idx.search_rows(Q, k)
    with idx.options(hi_search=0):
        S2, R2 = idx.search_rows(Q, k)
    assert np.array_equal(R1, R2) and _same(S1, S2) and not np.isin(R1, dead).any()
    few = np.zeros(n, bool)
    few[rng.choice(n, 37, replace=False)] = True
    S3, R3 = idx.search_rows(Q[0], k, chunk_filter=few)
    with idx.options(hi_search=0):
        S4, R4 = idx.search_rows(Q[0], k, chunk_filter=few)
    assert np.array_equal(R3, R4) and _same(S3, S4) and (R3 >= 0).sum() <= 37
    idx.close()


# ---- adversarial inputs for the single-vector bounds (round 4): the twins of tests/test_gpu_pp_pass.py's MaxSim cases ---------------------
@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("kind", ["aligned", "positive", "subnormal", "giant"])
@pytest.mark.parametrize("B", [1, 7, 128])
def test_row_searches_on_adversarial_data_equal_the_full_passes(kind, B, metric):
    """B <= 16: ranking on the HI plane (`search_rows_hi`); B >= 96: the fused top-k over the HI image (`search_rows_fused_hi`, candidate
    pass on the sixteen-group tile).  On corpora built to meet the error bound (residuals parallel to the query, no cancellation,
    subnormal hi halves, one giant row) both must return what the full-precision paths return: the same rows; the same score bits for the
    small batches (their candidates are re-scored by the very kernels of the full pass), and for the big ones the exact fp32 similarities,
    within a few ulps of the dense path's split-arithmetic sums."""
    import torch

    from tests.test_gpu_pp_pass import _adversarial

    raglite_amd.set_device(0)
    n, dim, k = 70_000, 1024, 40
    E, Q3 = _adversarial(torch, kind, n, dim, max(B, 2), 1)
    Q = Q3[:B, 0, :].contiguous()
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q if B > 1 else Q[0], k)
    st = idx.filter_stats()
    assert st["kind"] == ("rows_hi" if B <= 16 else "rows_fused_hi"), st
    with idx.options(hi_search=0, fused_topk=0):
        S0, R0 = idx.search_rows(Q if B > 1 else Q[0], k)
    assert idx.filter_stats()["kind"] == "none"
    if B <= 16:
        assert torch.equal(R, R0) and torch.equal(S.view(torch.int32), S0.view(torch.int32)), (kind, B, metric, st)
    else:
        scale = float(S0.abs().max())
        assert float((S - S0).abs().max()) <= 4e-6 * scale  # exact fp32 similarities vs the dense path's split-arithmetic sums
        differ = R != R0  # rows may swap only where the two arithmetics order near-equal scores differently
        assert float(differ.float().mean()) <= 0.02, float(differ.float().mean())
        assert torch.equal(torch.sort(R, dim=1).values[~differ.any(dim=1)], torch.sort(R0, dim=1).values[~differ.any(dim=1)])
    print(f"[{kind} B={B} {metric}] candidates per query mean {st['candidates_per_query_mean']:.0f} max {st['candidates_per_query_max']}, fallback {st['fallback']}")
    idx.close()


# ---- round 4: the unfiltered search lists its candidates from the selection's final kernel (select.hip: HiEmit), the filtered one still
# collects them with a pass over the scores (hi_filter.hip: collect_above_kernel) -- the two must see the same candidates ----------------------
@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("B", [1, 5, 16])
def test_candidates_listed_by_the_selection_equal_the_collected_ones(metric, B):
    n, dim, k = 70_000, 1024, 100
    E = oracle.synth_matrix(9960, n, dim)
    Q = oracle.synth_matrix(9961 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    st = idx.filter_stats()
    everything = np.ones(n, bool)
    Sf, Rf = idx.search_rows(Q, k, chunk_filter=everything)  # same rows eligible, through the collecting flow
    stf = idx.filter_stats()
    assert st["kind"] == stf["kind"] == "rows_hi" and not st["fallback"] and not stf["fallback"]
    assert st["candidates_per_query_max"] == stf["candidates_per_query_max"]
    assert abs(st["candidates_per_query_mean"] - stf["candidates_per_query_mean"]) < 1e-9
    assert k <= st["candidates_per_query_max"] < 1024
    assert np.array_equal(R, Rf) and _same(S, Sf)
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q, k)
    assert np.array_equal(R, R0) and _same(S, S0)
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_one_launch_fallback_selection_on_massive_ties(metric):
    """Every list overflows (3 000 copies of the best row per query) AND the scores tie massively: the guarded pass + the one-block
    selection (select.hip: guarded_select_kernel -> refine_in_bin) must return what the three-launch selection of the full path returns."""
    rng = np.random.default_rng(5)
    n, dim, k = 70_000, 1024, 100
    E = oracle.synth_matrix(9970, n, dim, "small_int")
    Q = oracle.synth_matrix(9971, 3, dim, "small_int")
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


# ---- l2 on the half-bytes route (round 6): `vector_search_distance_metric = "l2"` (`/root/reference/src/raglite/_config.py:69`, `_typing.py:123-134`:
# dist = |e - q|, sim = 1 - dist).  The approximate similarity comes from |e|^2 + |q|^2 - 2 e_hi.q, the bound lives on the squared distance, and the
# candidates are scored by the scan that sums (e - q)^2 directly -- the full-precision route's own kernel for up to four queries: the same bits.
@pytest.mark.parametrize("n,dim,B,k", [(80_000, 1024, 1, 100), (140_000, 512, 2, 10), (100_000, 1024, 4, 128), (530_000, 128, 3, 100), (70_000, 1536, 1, 50),
                                        (80_000, 3072, 2, 100), (400_000, 256, 2, 256), (800_000, 128, 1, 512)])
def test_l2_hi_search_equals_full_pass_bitwise(n, dim, B, k):  # (where the pivot route applies: k <= 128, >= 3 k group maxima of >= 1024 rows)
    E = oracle.synth_matrix(9500 + dim, n, dim)
    Q = oracle.synth_matrix(9600 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric="l2")
    S, R = idx.search_rows(Q if B > 1 else Q[0], k)
    st = idx.filter_stats()
    assert st["kind"] == "rows_hi" and not st["fallback"], st
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q if B > 1 else Q[0], k)
    assert np.array_equal(R, R0) and _same(S, S0)
    S, R = np.atleast_2d(S), np.atleast_2d(R)
    for b in (0, B - 1):
        sims = oracle.similarity(E, Q[b], "l2")
        assert_topk_close(S[b], R[b], sims, k, 4e-6 * max(1.0, float(np.abs(sims).max())))
    idx.close()


def test_l2_hi_search_integer_near_duplicates_filter_and_what_the_route_declines():
    rng = np.random.default_rng(12)
    n, dim, k = 80_000, 1024, 64
    E = oracle.synth_matrix(9700, n, dim, "small_int")
    Q = oracle.synth_matrix(9701, 3, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric="l2")
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["kind"] == "rows_hi"
    for b in range(3):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], "l2"), k)
        assert np.array_equal(R[b], ei) and _same(S[b], es.astype(np.float32))
    # a metadata filter and tombstones: the ranked flow under a row mask
    ok = rng.random(n) < 0.4
    S1, R1 = idx.search_rows(Q, k, chunk_filter=ok)
    assert idx.filter_stats()["kind"] == "rows_hi" and ok[R1].all()
    with idx.options(hi_search=0):
        S2, R2 = idx.search_rows(Q, k, chunk_filter=ok)
    assert np.array_equal(R1, R2) and _same(S1, S2)
    S1b, R1b = idx.search_rows(Q[:2], 100, chunk_filter=ok)  # (k > 64 under a mask: still enough group maxima at 80 000 rows)
    with idx.options(hi_search=0):
        S2b, R2b = idx.search_rows(Q[:2], 100, chunk_filter=ok)
    assert np.array_equal(R1b, R2b) and _same(S1b, S2b)
    # too few group maxima for the pivot at this k (l2 similarities crowd into one bin of the radix selection: only the pivot selects them
    # fast), more than four queries: the full-precision route (same results, the route says so)
    S3, R3 = idx.search_rows(Q[0], 300)
    assert idx.filter_stats()["kind"] != "rows_hi"
    es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[0], "l2"), 300)
    assert np.array_equal(R3, ei) and _same(S3, es.astype(np.float32))
    Q5 = oracle.synth_matrix(9702, 5, dim, "small_int")
    S4, R4 = idx.search_rows(Q5, 10)
    assert idx.filter_stats()["kind"] != "rows_hi"
    idx.close()
    # the reference's nearest neighbour IS often a near-duplicate: 5000 rows within 1e-4 of the query -- distances of ~3e-3 where |e|^2 + |q|^2 -
    # 2 e.q has lost every digit; the band holds more rows than a list, the guarded scan answers with the full route's bits
    E = oracle.synth_matrix(9800, n, dim)
    q = oracle.synth_matrix(9801, 1, dim)[0]
    dup = rng.choice(n, 5000, replace=False)
    E[dup] = (q[None, :] + 1e-4 * rng.standard_normal((5000, dim))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric="l2")
    S, R = idx.search_rows(q, 100)
    assert idx.filter_stats()["kind"] == "rows_hi" and idx.filter_stats()["fallback"]
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(q, 100)
    assert np.array_equal(R, R0) and _same(S, S0) and np.isin(R, dup).all()
    # ... and a few dozen near-duplicates only: the candidates hold them, no fallback, the same bits
    E2 = oracle.synth_matrix(9802, n, dim)
    few = rng.choice(n, 40, replace=False)
    E2[few] = (q[None, :] + 1e-4 * rng.standard_normal((40, dim))).astype(np.float32)
    idx2 = raglite_amd.DeviceIndex(E2, metric="l2")
    S, R = idx2.search_rows(q, 100)
    assert idx2.filter_stats()["kind"] == "rows_hi" and not idx2.filter_stats()["fallback"]
    with idx2.options(hi_search=0):
        S0, R0 = idx2.search_rows(q, 100)
    assert np.array_equal(R, R0) and _same(S, S0) and np.isin(few, R).all()
    idx.close()
    idx2.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("B,k", [(1, 5), (3, 1), (1, 101), (3, 33), (2, 7)])
def test_odd_and_even_scratch_layouts_same_bits_and_the_route_answers(metric, B, k):
    """The gather buffer of the half-bytes search sits behind 2 nb k words of other scratch: with nb k odd it used to be 8-byte misaligned -- the
    stream kernel then declined AFTER the approximate pass (the full route answered: right results, both routes paid), and the l2 scan fell back
    to scalar loads (another order of summation: the last bit of parity with the full pass).  Found by scripts/soak_pivot.py (round 6)."""
    n, dim = 300_000, 256
    E = oracle.synth_matrix(9990, n, dim)
    Q = oracle.synth_matrix(9991, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    q = Q if B > 1 else Q[0]
    S, R = idx.search_rows(q, k)
    st = idx.filter_stats()
    assert st["kind"] == "rows_hi" and not st["fallback"], st
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(q, k)
        assert idx.filter_stats()["kind"] != "rows_hi"
    assert np.array_equal(R, R0) and _same(S, S0)
    idx.close()
