"""Embedders wider than bge-m3 (round 6): the reference takes any litellm embedder (`/root/reference/src/raglite/_embed.py:155-158`:
1536-, 3072-wide models; 2048 / 2560 / 4096 exist), and the half-bytes routes used to stop at dim 1024 -- a wider index ran everything on its
full-precision paths.  Now for dim % 128 == 0 up to 4096:

* the exact re-scoring kernel (`maxsim_generic.hip: maxsim_pairs_wide_kernel`, wave-private 128-column windows of the query) -- the bits of
  `maxsim_pairs_kernel`'s arithmetic: a 1024-wide problem padded with zero columns to 1152 gives the same bits as the 1024-wide kernel, integer
  data equals the oracle, float data the float64 oracle within 2e-6 of the score scale;
* the bound-filtered MaxSim batch (`rl_maxsim_topk_batch`: HI image + sixteen-query pass + exact re-scoring; score =
  sum_i max_{j in chunk} Q[i].D[j], `_search.py:143-149` behind the reranker call :394-396) on 1536 / 2048 / 3072-wide indexes: route taken,
  same chunks as the full-precision passes, the oracle's scores; the bound's rounding term grows with dim (api.hip: sum_eps);
* the fused exact row top-k of a big batch (`_search.py:69-79` at B >= 96) over a 1536-wide index."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, ragged_offsets

pytestmark = pytest.mark.gpu


def _long_offsets(rng, n):
    sizes = [1000]
    while sum(sizes) < n:
        sizes.append(int(rng.integers(1, 101)))
    off = np.concatenate(([0], np.cumsum(sizes)))
    off = off[off <= n]
    return (off if off[-1] == n else np.concatenate((off, [n]))).astype(np.int64)


@pytest.mark.parametrize("layout", ["ragged", "one_row", "long", "with_empty"])
@pytest.mark.parametrize("nq,n_queries,n_cand", [(32, 5, 100), (17, 40, 37), (1, 3, 700), (32, 1, 1)])
def test_wide_kernel_on_zero_padded_columns_is_the_1024_kernel_bit_for_bit(layout, nq, n_queries, n_cand):
    rng = np.random.default_rng(nq + n_cand)
    n, dim = 5_000, 1024
    off = {"ragged": lambda: ragged_offsets(rng, n, 1, 15), "one_row": lambda: np.arange(n + 1, dtype=np.int64),
           "long": lambda: _long_offsets(rng, n), "with_empty": lambda: ragged_offsets(rng, n, 1, 15, empty_every=7)}[layout]()
    E = oracle.synth_matrix(21_000, n, dim)
    Q = np.stack([oracle.synth_matrix(21_100 + i, nq, dim) for i in range(n_queries)])
    cand = rng.integers(0, len(off) - 1, (n_queries, n_cand)).astype(np.int32)
    cand[rng.random(cand.shape) < 0.05] = -1
    narrow = raglite_amd.DeviceIndex(E, off, metric="dot")
    want = narrow.maxsim_rerank(Q, cand)
    narrow.close()
    for pad in (128, 512):  # 1152 = 9 windows, 1536 = 12
        Ep = np.concatenate((E, np.zeros((n, pad), np.float32)), axis=1)
        Qp = np.concatenate((Q, np.zeros((n_queries, nq, pad), np.float32)), axis=2)
        wide = raglite_amd.DeviceIndex(Ep, off, metric="dot")
        got = wide.maxsim_rerank(Qp, cand)
        wide.close()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), pad


@pytest.mark.parametrize("dim", [1152, 1536, 2048, 3072, 4096])
def test_wide_kernel_integer_data_is_the_oracle_and_float_data_close(dim):
    rng = np.random.default_rng(dim)
    n, nq = 3_000, 32
    off = ragged_offsets(rng, n, 1, 15)
    cand = rng.integers(0, len(off) - 1, (3, 200)).astype(np.int32)
    for kind, exact in (("small_int", True), ("uniform", False)):
        E = oracle.synth_matrix(21_500 + dim, n, dim, kind)
        Q = np.stack([oracle.synth_matrix(21_600 + i, nq, dim, kind) for i in range(3)])
        idx = raglite_amd.DeviceIndex(E, off, metric="dot")
        got = idx.maxsim_rerank(Q, cand)
        idx.close()
        for b in range(3):
            ref = oracle.maxsim_scores(E, off, Q[b], np.float64)[cand[b]]
            if exact:
                assert np.array_equal(got[b].astype(np.float64), ref)
            else:
                np.testing.assert_allclose(got[b], ref, rtol=0, atol=2e-6 * float(np.abs(ref).max()))


@pytest.mark.parametrize("dim,n", [(1536, 48_000), (2048, 36_000), (3072, 24_000)])
def test_maxsim_batch_over_a_wide_index_takes_the_bound_filtered_route(dim, n):
    rng = np.random.default_rng(dim + 1)
    off = ragged_offsets(rng, n, 1, 15)
    k, n_queries = 100, 9
    # integer data: bit-identical to the oracle, ties included
    E = oracle.synth_matrix(22_000 + dim, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(22_100 + i, 32, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"], st
    for i in (0, 4, 8):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc) and np.array_equal(bs[i], ws), i
    idx.close()
    # float data: the chunks of the full-precision passes, the float64 oracle's scores
    E = oracle.synth_matrix(22_200 + dim, n, dim)
    Qb = np.stack([oracle.synth_matrix(22_300 + i, 32, dim) for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"], st
    assert st["candidates_per_query_max"] < 2048
    with idx.options(hi_maxsim=0):
        fs, fc = idx.maxsim_topk_batch(Qb, k)
    for i in range(n_queries):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        tol = 2e-6 * float(np.abs(ref).max())
        assert_topk_close(bs[i], bc[i], ref, k, tol)
        assert set(bc[i].tolist()) == set(fc[i].tolist())
    # fp16 queries through the same route
    bs16, bc16 = idx.maxsim_topk_batch(Qb.astype(np.float16), k)
    for i in (0, 8):
        ref = oracle.maxsim_scores(E, off, Qb[i].astype(np.float16).astype(np.float32), np.float64)
        assert_topk_close(bs16[i], bc16[i], ref, k, 2e-6 * float(np.abs(ref).max()))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_fused_row_topk_of_a_big_batch_over_a_1536_wide_index(metric):
    n, dim, B, k = 48_000, 1536, 128, 50
    E = oracle.synth_matrix(23_000, n, dim)
    Q = oracle.synth_matrix(23_100, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    s, r = idx.search_rows(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "rows_fused_hi" and not st["fallback"], st
    for b in (0, 63, 127):
        tol = 2e-6 * (1.0 if metric == "cosine" else max(1.0, float(np.linalg.norm(E, axis=1).max() * np.linalg.norm(Q[b]))))
        assert_topk_close(s[b], r[b], oracle.similarity(E, Q[b], metric), k, tol)
    with idx.options(fused_hi=0):  # (the fused top-k over the pre-split image: three products per multiply, what a wide index ran before)
        s0, r0 = idx.search_rows(Q, k)
    for b in (0, 127):
        assert len(set(r[b].tolist()) ^ set(r0[b].tolist())) <= 2  # (the two routes' last bits differ: a swap at the k-th place is allowed)
    idx.close()


# ---- a few queries at a time over a wide index: the half-bytes row search through the packed scan (api.hip: search_rows_hi, `wide`) ------------
@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("n,dim,B,k", [(48_000, 1536, 1, 100), (70_000, 1536, 4, 512), (36_000, 2048, 3, 10), (24_000, 3072, 2, 100), (70_000, 4096, 1, 50)])
def test_few_queries_row_search_over_a_wide_index(metric, n, dim, B, k):
    E = oracle.synth_matrix(24_000 + dim, n, dim)
    Q = oracle.synth_matrix(24_100 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q if B > 1 else Q[0], k)
    st = idx.filter_stats()
    want_route = n >= 65_536  # (the route's own size gate)
    assert (st["kind"] == "rows_hi") == want_route, st
    assert not want_route or not st["fallback"]
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q if B > 1 else Q[0], k)
    S, R, S0, R0 = np.atleast_2d(S), np.atleast_2d(R), np.atleast_2d(S0), np.atleast_2d(R0)
    for b in range(B):
        sims = oracle.similarity(E, Q[b], metric)
        assert_topk_close(S[b], R[b], sims, k, 2e-6 * max(1.0, float(np.abs(sims).max())))
        if metric == "dot":  # (1 + e.q by the same scan: the same bits; a cosine's query norm is summed in another order by the two routes)
            assert np.array_equal(R[b], R0[b]) and np.array_equal(S[b].view(np.uint32), S0[b].view(np.uint32))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_wide_row_search_integer_data_filter_tombstones_and_the_guarded_full_pass(metric):
    rng = np.random.default_rng(31)
    n, dim, k = 70_000, 1536, 64
    E = oracle.synth_matrix(25_000, n, dim, "small_int")
    Q = oracle.synth_matrix(25_001, 3, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["kind"] == "rows_hi"
    from tests.util import sim_fp32_exact
    for b in range(3):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei) and np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    ok = rng.random(n) < 0.4
    S1, R1 = idx.search_rows(Q, k, chunk_filter=ok)
    assert ok[R1].all()
    for b in range(3):
        sims = np.where(ok, sim_fp32_exact(E, Q[b], metric), -np.inf)
        es, ei = oracle.topk_desc(sims, k)
        assert np.array_equal(R1[b], ei)
    idx.close()
    # near-duplicates: more rows inside the band than a list holds -> the flag -> the guarded fp32 scan + selection answer
    E = oracle.synth_matrix(25_100, n, dim)
    q = oracle.synth_matrix(25_101, 1, dim)[0]
    dup = rng.choice(n, 5000, replace=False)
    E[dup] = (q[None, :] * 0.9 + 1e-4 * rng.standard_normal((5000, dim))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(q, 100)
    assert idx.filter_stats()["fallback"]
    assert np.isin(R, dup).all()
    sims = oracle.similarity(E, q, metric)
    assert_topk_close(S, R, sims, 100, 2e-6 * max(1.0, float(np.abs(sims).max())))
    idx.close()


@pytest.mark.parametrize("dim,n", [(1536, 48_000), (3072, 24_000)])
def test_one_maxsim_query_over_a_wide_index_goes_through_the_pass(dim, n):
    """`rl_maxsim_topk` (one user query at a time: how the reference calls the reranker, `_search.py:394-396`) and batches of one or two on a wide
    index: no streaming kernel covers dim > 1024, so even one query takes the bound-filtered pipeline (api.hip: gemm_min_queries)."""
    rng = np.random.default_rng(dim + 7)
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(26_000 + dim, n, dim)
    Qb = np.stack([oracle.synth_matrix(26_100 + i, 32, dim) for i in range(2)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s1, c1 = idx.maxsim_topk(Qb[0], 100)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"], st
    ref = oracle.maxsim_scores(E, off, Qb[0], np.float64)
    assert_topk_close(s1, c1, ref, 100, 2e-6 * float(np.abs(ref).max()))
    s2, c2 = idx.maxsim_topk_batch(Qb, 100)
    assert np.array_equal(c2[0], c1) and np.array_equal(s2[0].view(np.uint32), s1.view(np.uint32))
    ref = oracle.maxsim_scores(E, off, Qb[1], np.float64)
    assert_topk_close(s2[1], c2[1], ref, 100, 2e-6 * float(np.abs(ref).max()))
    dead = np.unique(c1[:7])
    idx.delete_chunks(dead)
    s3, c3 = idx.maxsim_topk(Qb[0], 100)
    assert not np.isin(c3, dead).any()
    ok = rng.random(len(off) - 1) < 0.5  # (a metadata filter rides on the same pipeline: filtered-out chunks rank -inf in the approximate scores)
    s4, c4 = idx.maxsim_topk(Qb[0], 50, chunk_filter=ok)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"], st
    assert ok[c4].all() and not np.isin(c4, dead).any()
    ref = oracle.maxsim_scores(E, off, Qb[0], np.float64)
    ref[~ok] = -np.inf
    ref[dead] = -np.inf
    assert_topk_close(s4, c4, ref, 50, 2e-6 * float(np.abs(ref[np.isfinite(ref)]).max()))
    tiny = np.zeros(len(off) - 1, bool)  # fewer eligible chunks than k: never a masked chunk in the result (the guarded pass answers, padded)
    tiny[rng.choice(len(off) - 1, 9, replace=False)] = True
    tiny[dead] = False
    s5, c5 = idx.maxsim_topk(Qb[0], 50, chunk_filter=tiny)
    kk = int(tiny.sum())
    assert tiny[c5[:kk]].all() and (c5[kk:] == -1).all() and np.isneginf(s5[kk:]).all() and idx.filter_stats()["fallback"]
    idx.close()


def test_wide_index_keeps_rows_and_hi_image_only_and_its_guarded_fallback_scores_every_chunk_exactly():
    """A wide index runs the batch on rows + HI image (1.5 x the corpus) like a 1024-wide one; what stands behind the flag there is the exact
    re-scoring kernel over EVERY chunk (no streaming kernel covers dim > 1024).  4 000 near-identical one-row chunks defeat the bound: the
    first batch falls back through it, leaves word, and the second asks for the pre-split image (the eight-query full-precision pass)."""
    rng = np.random.default_rng(16)
    n, dim = 48_000, 1536
    off = np.arange(n + 1, dtype=np.int64)
    E = oracle.synth_matrix(27_000, n, dim)
    Qb = np.stack([oracle.synth_matrix(27_100 + i, 8, dim) for i in range(4)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s0, c0 = idx.maxsim_topk_batch(Qb, 100)
    mem = idx.memory()
    assert not idx.filter_stats()["fallback"] and mem["presplit_image"] == 0 and mem["hi_image"] > 0
    idx.close()
    hot = rng.choice(n, 4000, replace=False)
    E[hot] = (3.0 * Qb[:, 0].sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, dim))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s1, c1 = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"] and idx.memory()["presplit_image"] == 0
    s2, c2 = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"] and idx.memory()["presplit_image"] > 0
    for s_, c_ in ((s1, c1), (s2, c2)):
        for i in range(len(Qb)):
            ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
            assert_topk_close(s_[i], c_[i], ref, 100, 2e-6 * float(np.abs(ref).max()))
            assert np.isin(c_[i], hot).all()
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("B", [5, 16, 50])
def test_mid_size_batches_over_a_wide_index_take_the_gemm_shaped_routes(metric, B):
    """Between the few-queries search (<= 4) and the big batches a 1024-wide index has the 32-queries-per-pass streaming kernel; a wide index had
    only its scan, ONE query per pass (16 queries over 650 k x 1536: 14 ms).  There the GEMM-shaped routes -- the fused top-k over the HI image,
    the dense GEMM for l2 / masks -- start at five queries (api.hip: rows_gemm_min)."""
    n, dim, k = 48_000, 1536, 40
    E = oracle.synth_matrix(28_000, n, dim)
    Q = oracle.synth_matrix(28_100 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    s, r = idx.search_rows(Q, k)
    st = idx.filter_stats()
    if metric != "l2":
        assert st["kind"] == "rows_fused_hi" and not st["fallback"], st
    for b in (0, B - 1):
        sims = oracle.similarity(E, Q[b], metric)
        scale = 1.0 if metric == "cosine" else max(1.0, float(np.linalg.norm(E, axis=1).max() * np.linalg.norm(Q[b])))
        assert_topk_close(s[b], r[b], sims, k, 4e-6 * scale)
    ok = np.random.default_rng(B).random(n) < 0.5  # a row mask: the dense route
    s1, r1 = idx.search_rows(Q, k, chunk_filter=ok)
    assert ok[r1].all()
    sims = np.where(ok, oracle.similarity(E, Q[0], metric), -np.inf)
    scale = 1.0 if metric == "cosine" else max(1.0, float(np.linalg.norm(E, axis=1).max() * np.linalg.norm(Q[0])))
    assert_topk_close(s1[0], r1[0], sims, k, 4e-6 * scale)
    idx.close()


# ---- fp16-STORED wide indexes (round 6): the reference's own storage precision (`_embed.py:140`, `_database.py:279-283`) at dim 1536 / 3072 ----------
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("dim", [1536, 3072])
def test_f16_stored_wide_index_row_searches_and_lifecycle(metric, dim):
    rng = np.random.default_rng(dim)
    n = 6_000
    off = ragged_offsets(rng, n, 1, 9)
    E16 = oracle.synth_matrix(29_000 + dim, n, dim).astype(np.float16)
    Ev = E16.astype(np.float32)
    idx = raglite_amd.DeviceIndex(E16, off, metric=metric, storage="f16")
    for B in (1, 3, 4, 7, 20):  # the packed scan (<= 4 per pass), the GEMM-shaped routes from five up
        Q = oracle.synth_matrix(29_100 + B, B, dim)
        S, R = idx.search_rows(Q, 40)
        for b in (0, B - 1):
            sims = oracle.similarity(Ev, Q[b], metric)
            scale = 1.0 if metric == "cosine" else max(1.0, float(np.linalg.norm(Ev, axis=1).max() * np.linalg.norm(Q[b])))
            assert_topk_close(S[b], R[b], sims, 40, 4e-6 * scale)
    n_chunks = len(off) - 1
    r2c = np.repeat(np.arange(n_chunks), np.diff(off))
    q = oracle.synth_matrix(29_200, 1, dim)[0]
    flt = rng.random(n_chunks) < 0.4
    s, r = idx.search_rows(q, 30, chunk_filter=flt)
    assert flt[r2c[r]].all()
    idx.delete_chunks(np.nonzero(~flt)[0])
    s2, r2 = idx.search_rows(q, 30)
    assert np.array_equal(r2, r) and np.array_equal(s2.view(np.uint32), s.view(np.uint32))
    extra = oracle.synth_matrix(29_300, 17, dim)
    idx.append(extra)
    s3, r3 = idx.search_rows(extra[5].astype(np.float16).astype(np.float32), 3)
    assert r3[0] == n + 5
    cs, cc, cn = idx.search_chunks(q, 40, 5)
    assert cn == 5 and flt[cc[cc < n_chunks]].all()
    idx.close()
    # integer data is exact in fp16: the oracle's rows and bits
    Ei = oracle.synth_matrix(29_400 + dim, n, dim, "small_int")
    Qi = oracle.synth_matrix(29_500, 3, dim, "small_int")
    idx = raglite_amd.DeviceIndex(Ei.astype(np.float16), off, metric=metric, storage="f16")
    S, R = idx.search_rows(Qi, 25)
    from tests.util import sim_fp32_exact
    for b in range(3):
        es, ei = oracle.topk_desc(sim_fp32_exact(Ei, Qi[b], metric), 25)
        assert np.array_equal(R[b], ei) and np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    idx.close()


@pytest.mark.parametrize("dim,n", [(1536, 48_000), (3072, 24_000)])
def test_f16_stored_wide_index_maxsim(dim, n):
    rng = np.random.default_rng(dim + 3)
    off = ragged_offsets(rng, n, 1, 15)
    E16 = oracle.synth_matrix(29_600 + dim, n, dim).astype(np.float16)
    Ev = E16.astype(np.float32)
    Qb = np.stack([oracle.synth_matrix(29_700 + i, 32, dim) for i in range(9)])
    idx = raglite_amd.DeviceIndex(E16, off, metric="dot", storage="f16")
    assert idx.memory()["rows"] == n * dim * 2
    k = 100
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"], st
    for i in (0, 4, 8):
        ref = oracle.maxsim_scores(Ev, off, Qb[i], np.float64)
        assert_topk_close(bs[i], bc[i], ref, k, 2e-6 * float(np.abs(ref).max()))
    # fp16 queries over fp16 rows: the pass IS the score
    bs16, bc16 = idx.maxsim_topk_batch(Qb.astype(np.float16), k)
    assert idx.filter_stats()["kind"] == "maxsim_batch_f16_exact"
    ref = oracle.maxsim_scores(Ev, off, Qb[2].astype(np.float16).astype(np.float32), np.float64)
    assert_topk_close(bs16[2], bc16[2], ref, k, 2e-6 * float(np.abs(ref).max()))
    # one query; every chunk's score; the rerank of a candidate list
    s1, c1 = idx.maxsim_topk(Qb[0], k)
    assert np.array_equal(c1, bc[0])
    all_scores = idx.maxsim_scores(Qb[1])
    ref = oracle.maxsim_scores(Ev, off, Qb[1], np.float64)
    np.testing.assert_allclose(all_scores, ref, rtol=0, atol=2e-6 * float(np.abs(ref).max()))
    cand = rng.integers(0, len(off) - 1, (9, 60)).astype(np.int32)
    cand[:, ::7] = -1
    got = idx.maxsim_rerank(Qb, cand)
    for i in (0, 8):
        ref = oracle.maxsim_scores(Ev, off, Qb[i], np.float64)[np.maximum(cand[i], 0)]
        ok = cand[i] >= 0
        assert np.isneginf(got[i][~ok]).all()
        np.testing.assert_allclose(got[i][ok], ref[ok], rtol=0, atol=2e-6 * float(np.abs(ref).max()))
    idx.close()
    # a small wide fp16 index (no image): the exact kernel over every chunk, with empty chunks
    n2 = 3_000
    off2 = ragged_offsets(rng, n2, 1, 15, empty_every=9)
    Es = oracle.synth_matrix(29_800, n2, dim, "small_int")
    Qs = oracle.synth_matrix(29_801, 17, dim, "small_int")
    small = raglite_amd.DeviceIndex(Es.astype(np.float16), off2, metric="dot", storage="f16")
    s, c = small.maxsim_topk(Qs, 20)
    ws, wc = oracle.maxsim_topk(Es, off2, Qs, 20, np.float32)
    assert np.array_equal(c, wc) and np.array_equal(s, ws)
    small.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_small_wide_index_every_batch_size(metric, storage):
    """A wide index too small for any image (3 000 x 1536): 1 .. 130 queries through whatever route takes them (the scans, the dense GEMM from
    five queries up) -- the oracle's rows, scores within tolerance; integer data exactly."""
    n, dim, k = 3_000, 1536, 20
    E = oracle.synth_matrix(31_000, n, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E.astype(np.float16) if storage == "f16" else E, metric=metric, storage=storage)
    from tests.util import sim_fp32_exact
    for B in (1, 4, 5, 16, 50, 130):
        Q = oracle.synth_matrix(31_100 + B, B, dim, "small_int")
        S, R = idx.search_rows(Q, k)
        for b in (0, B - 1):
            es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
            assert np.array_equal(R[b], ei), (B, b)
            assert np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32)), (B, b)
    idx.close()
