"""The one-launch selection (select.hip `topk_block_kernel`, option `topk_block`, round 6): the exact top-k of up to 262 144 scores per query
by ONE block per query -- LDS histogram of the key's top 11 bits, the threshold bin kept in LDS, a second radix level inside it, ranking by
counting -- instead of the histogram / filter / final launches.  `ORDER BY dist LIMIT k` (`/root/reference/src/raglite/_search.py:75-79`)
with the tie order SQL leaves open fixed to (score desc, index asc), NaN last: the same unique 64-bit keys as the three-launch route, so
every result must equal it BIT FOR BIT and equal `oracle.topk_desc` -- random data, crowded bins, massive ties (the slow exact paths), specials,
every alignment, k from 1 to 2048."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle

pytestmark = pytest.mark.gpu


def _both(x, k):
    """Three routes, one result: 0 = histogram / filter / final, 1 = the block route, 2 = the block route with the thread-maximum prefilter."""
    out = {}
    for route in (0, 1, 2):
        raglite_amd.set_default_option("topk_block", route)
        try:
            out[route] = raglite_amd.topk(x, k)
        finally:
            raglite_amd.set_default_option("topk_block", 2)
    s0, i0 = out[0]
    for route in (1, 2):
        s1, i1 = out[route]
        assert np.array_equal(i0, i1), f"block route {route} and three-launch route disagree on the ids"
        assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32)), route
    return out[2]


@pytest.mark.parametrize("n,k", [(1, 1), (3, 7), (63, 64), (4097, 100), (5000, 2048), (125_011, 100), (125_012, 512), (262_144, 1000), (262_143, 1)])
def test_random_scores_every_alignment(n, k):
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal(n).astype(np.float32)
    s, i = _both(x, k)
    es, ei = oracle.topk_desc(x, k)
    kk = min(n, k)
    assert np.array_equal(i[:kk], ei) and np.array_equal(s[:kk], es)
    assert (i[kk:] == -1).all() and np.isneginf(s[kk:]).all()


def test_crowded_bins_like_maxsim_scores_and_a_batch():
    """Scores ~ N(480, 34): a handful of 11-bit bins hold everything, the threshold bin thousands of keys (the second radix level decides)."""
    rng = np.random.default_rng(5)
    X = (480.0 + 34.0 * rng.standard_normal((37, 125_003))).astype(np.float32)
    S, I = _both(X, 100)
    for b in (0, 17, 36):
        es, ei = oracle.topk_desc(X[b], 100)
        assert np.array_equal(I[b], ei) and np.array_equal(S[b], es)
    # the threshold bin larger than the LDS buffer (8192 keys): the exact slow path over the global scores
    x = (700.0 + 1e-3 * rng.standard_normal(60_000)).astype(np.float32)
    s, i = _both(x, 300)
    es, ei = oracle.topk_desc(x, 300)
    assert np.array_equal(i, ei) and np.array_equal(s, es)


def test_massive_ties_and_specials():
    rng = np.random.default_rng(6)
    x = rng.integers(0, 3, size=50_000).astype(np.float32)  # three distinct values: ties in index order, far more than any list holds
    s, i = _both(x, 1000)
    es, ei = oracle.topk_desc(x, 1000)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    x = np.ones(20_000, dtype=np.float32)  # the reference's all-ones corpus (tests/test_split_chunks.py:28)
    s, i = _both(x, 2048)
    assert i.tolist() == list(range(2048))
    # > 1024 keys on one 22-bit prefix but distinct in the low bits
    x = (1.0 + np.arange(3000, dtype=np.float32) * 2.0 ** -23).astype(np.float32)
    s, i = _both(x, 1500)
    es, ei = oracle.topk_desc(x, 1500)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    x = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0, np.nan, -1.0, 1.0], dtype=np.float32)
    s, i = _both(x, 9)
    assert i.tolist()[:3] == [3, 5, 8] and i.tolist()[-3:] == [4, 2, 6]
    assert np.isnan(s[-1]) and np.isnan(s[-2]) and np.isneginf(s[-3])
    x = np.full(5000, -np.inf, dtype=np.float32)
    x[[7, 4999]] = [2.0, 3.0]
    s, i = _both(x, 10)
    assert i[:2].tolist() == [4999, 7]


def test_device_tensors_unaligned_rows_and_the_pipelines_that_use_it():
    import torch

    rng = np.random.default_rng(7)
    X = torch.as_tensor(rng.standard_normal((5, 10_001)).astype(np.float32), device="cuda")  # ld = 10 001: scalar loads
    S, I = raglite_amd.topk(X, 50)
    for b in range(5):
        es, ei = oracle.topk_desc(X[b].cpu().numpy(), 50)
        assert np.array_equal(I[b].cpu().numpy(), ei) and np.array_equal(S[b].cpu().numpy(), es)
    # through an index: a MaxSim batch's approximate top-k and the single-query top-k run on it; integer data, bit-exact against the oracle
    from tests.util import ragged_offsets

    n, dim = 70_000, 1024
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(40_000, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(40_100 + i, 32, dim, "small_int") for i in range(5)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    for opt in (2, 1, 0):
        with idx.options(topk_block=opt):
            bs, bc = idx.maxsim_topk_batch(Qb, 100)
            ss, sc = idx.maxsim_topk(Qb[0], 100)
        for b in (0, 4):
            ws, wc = oracle.maxsim_topk(E, off, Qb[b], 100, np.float32)
            assert np.array_equal(bc[b], wc) and np.array_equal(bs[b], ws)
        assert np.array_equal(sc, bc[0]) and np.array_equal(ss, bs[0])
    idx.close()


def test_prefilter_worst_cases_fall_back_to_the_histogram_path():
    """The prefilter keeps the keys that reach the k-th largest of 1024 THREAD maxima (thread t reads the 16-byte groups t, t + 1024, ...).
    Data whose large scores all sit in the groups of a few threads makes that pivot useless -- more keys pass than the LDS buffer holds --
    and the block must notice and take the histogram path: same results as the other routes, bit for bit."""
    rng = np.random.default_rng(8)
    n = 200_000
    x = rng.standard_normal(n).astype(np.float32)
    g = np.arange(n) // 4  # 16-byte group of every element
    few = (g % 1024) < 40  # the groups of forty threads
    x[few] += 100.0        # ~7 800 large scores, all in forty threads' hands: the 100th thread maximum is an ordinary score
    for k in (100, 512):
        s, i = _both(x, k)
        es, ei = oracle.topk_desc(x, k)
        assert np.array_equal(i, ei) and np.array_equal(s, es)
    # sorted data, both directions (ascending: every thread's maximum sits in its last group)
    for y in (np.sort(x), np.sort(x)[::-1].copy()):
        s, i = _both(y, 300)
        es, ei = oracle.topk_desc(y, 300)
        assert np.array_equal(i, ei) and np.array_equal(s, es)
    # between 257 and 8192 survivors: the sorted path of the prefilter (k = 512 keeps ~700)
    z = (480.0 + 34.0 * rng.standard_normal(125_000)).astype(np.float32)
    s, i = _both(z, 512)
    es, ei = oracle.topk_desc(z, 512)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    # NaN-heavy: threads whose elements are all NaN have no maximum; fewer than k real scores altogether
    w = np.full(30_000, np.nan, dtype=np.float32)
    w[rng.choice(30_000, size=60, replace=False)] = rng.standard_normal(60).astype(np.float32)
    s, i = _both(w, 100)
    es, ei = oracle.topk_desc(w, 100)
    assert np.array_equal(i, ei) and np.array_equal(s.view(np.uint32)[:60], es.view(np.uint32)[:60]) and np.isnan(s[60:]).all()


def test_batches_over_more_scores_than_one_block_used_to_take():
    """A batch (>= 16 queries) over more than 262 144 scores per query also takes the block route when the prefilter is on (MaxSim chunk scores
    of a corpus of > 262 144 chunks crowd into one bin of the three-launch radix selection, whose exact slow path is one block reading the
    scores three more times): crowded scores, ties, an unaligned leading dimension, the prefilter's overflow -- equal to the three-launch route
    and the oracle bit for bit; fewer than 16 queries stay on the three-launch route and agree as well."""
    rng = np.random.default_rng(9)
    n = 700_003
    X = (755.0 + 3.0 * rng.standard_normal((16, n))).astype(np.float32)   # every score in ONE 11-bit bin
    X[3, rng.choice(n, size=5000, replace=False)] = 770.0                # massive ties on the threshold value
    g = np.arange(n) // 4
    X[5, (g % 1024) < 30] += 40.0                                        # the large scores in thirty threads' hands: prefilter overflow
    for k in (1, 100, 512, 600):
        S, I = _both(X, k)
        for b in (0, 3, 5, 15):
            es, ei = oracle.topk_desc(X[b], k)
            assert np.array_equal(I[b], ei) and np.array_equal(S[b].view(np.uint32), es.view(np.uint32)), (k, b)
    S, I = _both(X[:5], 100)
    for b in (0, 3):
        es, ei = oracle.topk_desc(X[b], 100)
        assert np.array_equal(I[b], ei) and np.array_equal(S[b], es)
