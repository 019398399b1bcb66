"""GPU parity of the half-bytes MaxSim batch (api.hip `rl_maxsim_topk_batch`, HI-image pipeline): batches of three or more
queries over a big fp32 corpus make their corpus passes over the HI halves of the fp16 split (ONE MFMA product per multiply
instead of three -- q_hi . e_hi; RAGLITE_HI_ONE_PRODUCT=0: two), bound every chunk's score error rigorously, and re-score the chunks that could be in the top-k exactly on
the fp32 matrix pipe (`maxsim_pairs_kernel`).

score[c] = sum_i max_{j in chunk c} Q[i].D[j] -- the multi-vector generalisation of
`/root/reference/src/raglite/_search.py:143-149` behind the reranker plugin call (:394-396).  Bars: integer data bit-identical
to the oracle (scores and chunk ordinals, ties included); float data: the same chunks as the full-precision passes
(RAGLITE_NO_HI_MAXSIM=1) and scores within 2e-6 of the score scale of the float64 oracle; corpora built to defeat the bound
(thousands of near-identical chunks) fall back on the device and still agree."""

import os

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, ragged_offsets

pytestmark = pytest.mark.gpu


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


N, DIM = 70_000, 1024  # >= 64 M elements: the index keeps a HI image


@pytest.mark.parametrize("nq,n_queries,k", [(32, 8, 50), (17, 11, 100), (1, 3, 10)])
def test_hi_maxsim_integer_bit_exact(nq, n_queries, k):
    rng = np.random.default_rng(nq + n_queries)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(10_000 + nq, N, DIM, "small_int")
    Qb = np.stack([oracle.synth_matrix(10_100 + i, nq, DIM, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in sorted({0, n_queries // 2, n_queries - 1}):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i], ws)
    idx.close()


def test_hi_maxsim_float_data_tombstones_and_switch():
    rng = np.random.default_rng(5)
    off = ragged_offsets(rng, N, 1, 15)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(10_200, N, DIM)
    Qb = np.stack([oracle.synth_matrix(10_300 + i, 32, DIM) for i in range(9)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    k = 100
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    with _env(RAGLITE_NO_HI_MAXSIM="1"):
        fs, fc = idx.maxsim_topk_batch(Qb, k)
    for i in range(9):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        tol = 2e-6 * float(np.abs(ref).max())
        assert_topk_close(bs[i], bc[i], ref, k, tol)
        assert set(bc[i].tolist()) == set(fc[i].tolist())  # (the two paths' scores differ in the last bits: order may too)
        np.testing.assert_allclose(np.sort(bs[i]), np.sort(fs[i]), rtol=0, atol=2 * tol)
    dead = np.unique(bc[:, :5])
    idx.delete_chunks(dead)
    bs2, bc2 = idx.maxsim_topk_batch(Qb, k)
    assert not np.isin(bc2, dead).any()
    for i in (0, 8):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64).copy()
        ref[dead] = -np.inf
        assert_topk_close(bs2[i], bc2[i], ref, k, 2e-6 * float(np.abs(ref[np.isfinite(ref)]).max()))
    assert n_chunks > k
    idx.close()


def test_near_identical_chunks_defeat_the_bound_and_the_full_passes_answer():
    """4 000 one-row chunks within 1e-4 of each other at the top of every query's ranking: more candidates than a list holds,
    the flag goes up, the guarded full-precision passes rank -- the result equals the run without the HI image."""
    rng = np.random.default_rng(6)
    off = np.arange(N + 1, dtype=np.int64)  # every row its own chunk
    E = oracle.synth_matrix(10_400, N, DIM)
    Qb = np.stack([oracle.synth_matrix(10_500 + i, 8, DIM) for i in range(4)])
    hot = rng.choice(N, 4000, replace=False)
    E[hot] = (3.0 * Qb[:, 0].sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, DIM))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    with _env(RAGLITE_NO_HI_MAXSIM="1"):
        fs, fc = idx.maxsim_topk_batch(Qb, 100)
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    assert np.isin(bc, hot).all()
    idx.close()


# ---- switches ----------------------------------------------------------------------------------------------------------------------
# RAGLITE_HI_ONE_PRODUCT (read per call; "1" is the default, "0" = two products: q_hi . e_hi + q_lo . e_hi through the eight-query
# kernel): the approximate pass multiplies q_hi . e_hi only -- a plain fp16 GEMM -- and the bound carries what the queries' hi halves
# drop.  RAGLITE_NO_PP=1: the one-product pass through the eight-query kernel instead of maxsim_pp.hip.  Results must not change.


@pytest.mark.parametrize("one,nopp", [("0", "0"), ("1", "1"), ("1", "0")])
def test_switches_integer_bit_exact(one, nopp):
    nq, n_queries, k = 32, 9, 100
    rng = np.random.default_rng(77)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(10_600, N, DIM, "small_int")
    Qb = np.stack([oracle.synth_matrix(10_700 + i, nq, DIM, "small_int") for i in range(n_queries)])
    with _env(RAGLITE_HI_ONE_PRODUCT=one, RAGLITE_NO_PP=nopp):
        idx = raglite_amd.DeviceIndex(E, off, metric="dot")
        bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in (0, 4, 8):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i], ws)
    idx.close()


@pytest.mark.parametrize("one,nopp", [("0", "0"), ("1", "1"), ("1", "0")])
def test_switches_float_data(one, nopp):
    rng = np.random.default_rng(78)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(10_800, N, DIM)
    Qb = np.stack([oracle.synth_matrix(10_900 + i, 32, DIM) for i in range(9)])
    k = 100
    with _env(RAGLITE_HI_ONE_PRODUCT=one, RAGLITE_NO_PP=nopp):
        idx = raglite_amd.DeviceIndex(E, off, metric="dot")
        bs, bc = idx.maxsim_topk_batch(Qb, k)
        s1, r1 = idx.search_rows(Qb[0, 0], 50)  # the single-query half-bytes search reads the same halves and norms
    with _env(RAGLITE_NO_HI_MAXSIM="1", RAGLITE_NO_HI_SEARCH="1"):
        fs, fc = idx.maxsim_topk_batch(Qb, k)
        s0, r0 = idx.search_rows(Qb[0, 0], 50)
    assert np.array_equal(r1, r0) and np.array_equal(s1.view(np.uint32), s0.view(np.uint32))
    for i in range(9):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        tol = 2e-6 * float(np.abs(ref).max())
        assert_topk_close(bs[i], bc[i], ref, k, tol)
        assert set(bc[i].tolist()) == set(fc[i].tolist())
    idx.close()


def test_one_product_fallback_on_near_identical_chunks():
    rng = np.random.default_rng(79)
    off = np.arange(N + 1, dtype=np.int64)
    E = oracle.synth_matrix(11_000, N, DIM)
    Qb = np.stack([oracle.synth_matrix(11_100 + i, 8, DIM) for i in range(4)])
    hot = rng.choice(N, 4000, replace=False)
    E[hot] = (3.0 * Qb[:, 0].sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, DIM))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"]
    with _env(RAGLITE_NO_HI_MAXSIM="1"):
        fs, fc = idx.maxsim_topk_batch(Qb, 100)
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    idx.close()
