"""The row-packing pairs kernel (`maxsim_generic.hip: maxsim_pairs_packed_kernel`, option `pairs_packed`, the default) against the kernel
that gives every candidate chunk its own 16-row MFMA tiles (`pairs_packed = 0`): the exact MaxSim of (query, candidate chunk) pairs --
score = sum_i max_{j in chunk} Q[i].D[j], `/root/reference/src/raglite/_search.py:143-149` behind the reranker plugin call (:394-396) --
must come out bit for bit the same (the k steps of a (row, query vector) pair accumulate in the same order whatever slot of a tile the row
sits in; max is exact; the sum over the query vectors runs the same tree), and equal to the oracle on integer data."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import ragged_offsets

pytestmark = pytest.mark.gpu


def _offsets(rng, layout, n):
    if layout == "ragged":
        return ragged_offsets(rng, n, 1, 15)
    if layout == "one_row":
        return np.arange(n + 1, dtype=np.int64)
    if layout == "long":  # chunks of 1 .. 100 rows and one of 1000: candidates that span many tiles, tiles that hold many candidates
        sizes = [1000]
        while sum(sizes) < n:
            sizes.append(int(rng.integers(1, 101)))
        off = np.concatenate(([0], np.cumsum(sizes)))
        off = off[off <= n]
        return (off if off[-1] == n else np.concatenate((off, [n]))).astype(np.int64)
    if layout == "with_empty":
        return ragged_offsets(rng, n, 1, 15, empty_every=7)
    raise AssertionError(layout)


@pytest.mark.parametrize("dim", [256, 384, 1024])
@pytest.mark.parametrize("layout", ["ragged", "one_row", "long", "with_empty"])
@pytest.mark.parametrize("nq,n_queries,n_cand", [(32, 5, 100), (17, 128, 37), (1, 3, 700), (32, 1, 1)])
def test_packed_equals_unpacked_bitwise(dim, layout, nq, n_queries, n_cand):
    rng = np.random.default_rng(dim + nq + n_cand)
    n = 6_000
    off = _offsets(rng, layout, n)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(13_000 + dim, n, dim)
    Q = np.stack([oracle.synth_matrix(13_100 + i, nq, dim) for i in range(n_queries)])
    cand = rng.integers(0, n_chunks, (n_queries, n_cand)).astype(np.int32)
    cand[rng.random(cand.shape) < 0.05] = -1  # the padding of a search result
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    packed = idx.maxsim_rerank(Q, cand)  # (the default, pairs_packed = 2: sixteen waves per workgroup, round 6)
    with idx.options(pairs_packed=0):
        plain = idx.maxsim_rerank(Q, cand)
    with idx.options(pairs_packed=1):
        eight = idx.maxsim_rerank(Q, cand)
    assert np.array_equal(packed.view(np.uint32), plain.view(np.uint32))
    assert np.array_equal(eight.view(np.uint32), plain.view(np.uint32))
    empty = np.diff(off)[np.maximum(cand, 0)] == 0
    assert np.isneginf(packed[(cand < 0) | empty]).all() and np.isfinite(packed[(cand >= 0) & ~empty]).all()
    idx.close()


def test_packed_integer_data_is_the_oracle():
    rng = np.random.default_rng(3)
    n, dim, nq = 5_000, 512, 32
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(13_500, n, dim, "small_int")
    Q = np.stack([oracle.synth_matrix(13_600 + i, nq, dim, "small_int") for i in range(4)])
    cand = rng.integers(0, len(off) - 1, (4, 250)).astype(np.int32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    got = idx.maxsim_rerank(Q, cand)
    for b in range(4):
        ref = oracle.maxsim_scores(E, off, Q[b], np.float64)[cand[b]]
        assert np.array_equal(got[b].astype(np.float64), ref)
    idx.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_batch_pipeline_same_bits_with_either_kernel(storage):
    """`rl_maxsim_topk_batch` re-scores its candidates with the pairs kernel (fp32 or fp16-stored rows): same results either way."""
    rng = np.random.default_rng(9)
    n, dim = 70_000, 1024
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(13_700, n, dim)
    if storage == "f16":
        E = E.astype(np.float16)
    Qb = np.stack([oracle.synth_matrix(13_800 + i, 32, dim) for i in range(19)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
    s1, c1 = idx.maxsim_topk_batch(Qb, 100)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"]
    for mode in (0, 1):
        with idx.options(pairs_packed=mode):
            s0, c0 = idx.maxsim_topk_batch(Qb, 100)
        assert np.array_equal(c0, c1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)), mode
    idx.close()
