"""One or two MaxSim queries over an fp32 index: the half-width route (api.hip `maxsim_few_hi_plane`, option `hi_few`, round 6).

How the reference calls the reranker -- one user query at a time (`/root/reference/src/raglite/_search.py:394-396`; the score is
`sum_i max_{j in chunk} q_i . d_j`, the multi-vector generalisation of `:143-149`).  Until round 5 one or two queries streamed the fp32
rows (4 B per element); now the approximate pass is the HBM-bound streaming kernel over the row-major fp16 HI plane (2 B per element, both
halves of the query multiplied: it drops e_lo only), then the batch pipeline's own stages: exact top-k of the approximate scores, the
rigorous bound m = (max|e_lo| + 2^-12 max|e|) sum_i |q_i|, the second threshold from the exact scores of the approximate top-k, exact
re-scoring of the candidates over the rows, ranking; the streaming pass over the rows behind a device flag.

Bars: integer-valued data bit-identical to the oracle (scores and chunk ordinals, ties included) and to the route it replaces
(`hi_few = 0`); float data the same chunks as `hi_few = 0`, scores within 2e-6 of the score scale of float64 (tie-aware top-k check);
tombstones; corpora built to defeat the bound fall back on the device and still agree; with and without the pre-split image."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, ragged_offsets

pytestmark = pytest.mark.gpu

N, DIM = 70_000, 1024  # >= 64 M elements: the index may keep a HI plane


@pytest.fixture(autouse=True, params=[1, 0], ids=["pivot", "ranked"])
def _both_candidate_routes(request):
    """Every case under hi_pivot = 1 (round 6: the candidates from a pivot over group maxima, no approximate ranking; k <= 128 and >= 3 k groups)
    and hi_pivot = 0 (top-k of the approximate scores + second threshold): one result."""
    raglite_amd.set_default_option("hi_pivot", request.param)
    yield
    raglite_amd.set_default_option("hi_pivot", 1)


def _route(idx):
    st = idx.filter_stats()
    return st["kind"], bool(st["fallback"])


@pytest.mark.parametrize("nq,k", [(32, 100), (17, 50), (16, 10), (1, 10)])
def test_single_query_integer_bit_exact(nq, k):
    rng = np.random.default_rng(nq)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(30_000 + nq, N, DIM, "small_int")
    Q = oracle.synth_matrix(30_100 + nq, nq, DIM, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s, c = idx.maxsim_topk(Q, k)
    assert _route(idx) == ("maxsim_batch_hi", False) and idx.memory()["hi_plane"] > 0
    ws, wc = oracle.maxsim_topk(E, off, Q, k, np.float32)
    assert np.array_equal(c, wc), (c[:8], wc[:8])
    assert np.array_equal(s, ws)
    with idx.options(hi_few=0):
        fs, fc = idx.maxsim_topk(Q, k)
        assert _route(idx)[0] == "none"
    assert np.array_equal(fc, c) and np.array_equal(fs, s)
    idx.close()


@pytest.mark.parametrize("n_queries,nq", [(1, 32), (2, 32), (2, 8), (10, 32)])
def test_batches_of_one_two_and_a_batch_with_two_left_over(n_queries, nq):
    """Batches of < 3 queries take the route whole; a batch of ten ranks eight through the sixteen-query pass and the two left over here."""
    rng = np.random.default_rng(100 + n_queries + nq)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(30_200 + nq, N, DIM, "small_int")
    Qb = np.stack([oracle.synth_matrix(30_300 + i, nq, DIM, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    k = 64
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    assert _route(idx) == ("maxsim_batch_hi", False)
    for i in range(n_queries):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i], ws)
    idx.close()


def test_float_data_device_tensors_tombstones_and_the_route_it_replaces():
    import torch

    rng = np.random.default_rng(7)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(30_400, N, DIM)
    Qb = np.stack([oracle.synth_matrix(30_500 + i, 32, DIM) for i in range(2)])
    idx = raglite_amd.DeviceIndex(torch.as_tensor(E, device="cuda"), off, metric="dot")
    k = 100
    Qd = torch.as_tensor(Qb, device="cuda")
    refs = [oracle.maxsim_scores(E, off, Qb[i], np.float64) for i in range(2)]
    for i in range(2):
        s, c = idx.maxsim_topk(Qd[i], k)
        assert _route(idx) == ("maxsim_batch_hi", False)
        with idx.options(hi_few=0):
            fs, fc = idx.maxsim_topk(Qd[i], k)
        s, c, fs, fc = (x.cpu().numpy() for x in (s, c, fs, fc))
        tol = 2e-6 * float(np.abs(refs[i]).max())
        assert_topk_close(s, c, refs[i], k, tol)
        assert set(c.tolist()) == set(fc.tolist())  # (the two routes' scores differ in the last bits: order may too)
        np.testing.assert_allclose(np.sort(s), np.sort(fs), rtol=0, atol=2 * tol)
    bs, bc = idx.maxsim_topk_batch(Qd, k)  # two queries of 32 vectors: ONE pass over the plane
    assert _route(idx) == ("maxsim_batch_hi", False)
    bs, bc = bs.cpu().numpy(), bc.cpu().numpy()
    for i in range(2):
        assert_topk_close(bs[i], bc[i], refs[i], k, 2e-6 * float(np.abs(refs[i]).max()))
    dead = np.unique(bc[:, :5])
    idx.delete_chunks(dead)
    s2, c2 = idx.maxsim_topk(Qd[0], k)
    s2, c2 = s2.cpu().numpy(), c2.cpu().numpy()
    assert not np.isin(c2, dead).any()
    ref = refs[0].copy()
    ref[dead] = -np.inf
    assert_topk_close(s2, c2, ref, k, 2e-6 * float(np.abs(ref[np.isfinite(ref)]).max()))
    idx.close()


@pytest.mark.parametrize("with_image", [False, True])
def test_near_identical_chunks_defeat_the_bound_and_the_rows_answer(with_image):
    """4 000 one-row chunks within 1e-4 of each other at the top of the ranking: more candidates than a list holds, the flag goes up, the
    guarded streaming pass over the rows + exact selection answer -- bit for bit what `hi_few = 0` returns (the same kernel over the same rows).
    With the pre-split image present the fallback must STILL stream the rows (this route laid out no query fragments for the eight-query pass)."""
    rng = np.random.default_rng(6)
    off = np.arange(N + 1, dtype=np.int64)
    E = oracle.synth_matrix(30_600, N, DIM)
    Q = oracle.synth_matrix(30_700, 8, DIM)
    hot = rng.choice(N, 4000, replace=False)
    E[hot] = (3.0 * Q.sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, DIM))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    if with_image:
        assert "presplit" in idx.prepare("presplit")
    s, c = idx.maxsim_topk(Q, 100)
    assert _route(idx) == ("maxsim_batch_hi", True)
    with idx.options(hi_few=0):
        fs, fc = idx.maxsim_topk(Q, 100)
    assert np.array_equal(c, fc) and np.array_equal(s.view(np.uint32), fs.view(np.uint32))
    assert np.isin(c, hot).all()
    idx.close()


def test_where_the_route_does_not_apply_the_rows_are_streamed_as_before():
    rng = np.random.default_rng(8)
    n = 20_000  # < 64 M elements: no HI plane
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(30_800, n, DIM, "small_int")
    Q = oracle.synth_matrix(30_900, 32, DIM, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s, c = idx.maxsim_topk(Q, 10)
    assert _route(idx)[0] == "none" and idx.memory()["hi_plane"] == 0
    ws, wc = oracle.maxsim_topk(E, off, Q, 10, np.float32)
    assert np.array_equal(c, wc) and np.array_equal(s, ws)
    idx.close()
    # a metadata filter (filter-first semantics, `_search.py:105-119`) rides on the half-width route since round 6: filtered-out chunks rank
    # -inf in the approximate scores, like tombstones; with the route off the masked full pass answers -- the same bits on integer data
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(30_801, N, DIM, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    mask = rng.random(len(off) - 1) < 0.5
    s, c = idx.maxsim_topk(Q, 10, chunk_filter=mask)
    assert _route(idx)[0] == "maxsim_batch_hi" and not _route(idx)[1]
    ws, wc = oracle.maxsim_topk_filtered(E, off, Q, 10, mask, np.float32)
    assert np.array_equal(c, wc) and np.array_equal(s, ws)
    with idx.options(hi_few=0):
        s0, c0 = idx.maxsim_topk(Q, 10, chunk_filter=mask)
    assert np.array_equal(c0, wc) and np.array_equal(s0, ws)
    # ... together with tombstones; a filter that leaves fewer chunks than k (the bound is unusable: the guarded pass answers, padded)
    dead = np.unique(wc[:4])
    idx.delete_chunks(dead)
    live = mask.copy()
    live[dead] = False
    s, c = idx.maxsim_topk(Q, 10, chunk_filter=mask)
    ws, wc = oracle.maxsim_topk_filtered(E, off, Q, 10, live, np.float32)
    assert np.array_equal(c, wc) and np.array_equal(s, ws)
    tiny = np.zeros(len(off) - 1, bool)
    tiny[rng.choice(len(off) - 1, 6, replace=False)] = True
    tiny[dead] = False
    s, c = idx.maxsim_topk(Q, 10, chunk_filter=tiny)
    ws, wc = oracle.maxsim_topk_filtered(E, off, Q, 10, tiny, np.float32)
    kk = int(tiny.sum())
    assert np.array_equal(c[:kk], wc[:kk]) and np.array_equal(s[:kk], ws[:kk]) and (c[kk:] == -1).all() and np.isneginf(s[kk:]).all()
    # float data: the float64 oracle's chunks among the eligible ones
    Ef = oracle.synth_matrix(30_802, N, DIM)
    Qf = oracle.synth_matrix(30_902, 32, DIM)
    fidx = raglite_amd.DeviceIndex(Ef, off, metric="dot")
    s, c = fidx.maxsim_topk(Qf, 100, chunk_filter=mask)
    assert _route(fidx)[0] == "maxsim_batch_hi" and mask[c].all()
    ref = oracle.maxsim_scores(Ef, off, Qf, np.float64)
    ref[~mask] = -np.inf
    from tests.util import assert_topk_close
    assert_topk_close(s, c, ref, 100, 2e-6 * float(np.abs(ref[np.isfinite(ref)]).max()))
    fidx.close()
    # fewer chunks than k: padding (-inf, -1) behind the real ones
    few = raglite_amd.DeviceIndex(E, np.array([0, N // 2, N], dtype=np.int64), metric="dot")
    s, c = few.maxsim_topk(Q, 5)
    ws, wc = oracle.maxsim_topk(E, np.array([0, N // 2, N], dtype=np.int64), Q, 5, np.float32)
    assert c[:2].tolist() == wc.tolist() and (c[2:] == -1).all() and np.array_equal(s[:2], ws) and np.isneginf(s[2:]).all()
    few.close()
    idx.close()
