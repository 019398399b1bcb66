"""Oracle self-consistency + known-answer values (CPU)."""

import numpy as np
import pytest

from oracle import oracle
from tests.util import ragged_offsets


def test_synth_known_answers():
    # Pinned values: any change to the generator breaks oracle <-> HIP bit-identity of test corpora.
    u = oracle.synth_uniform(2, 0, 4)
    assert u.dtype == np.float32 and np.all((u >= -1) & (u < 1))
    bits = oracle.synth_bits(2, 0, 2)
    assert bits.dtype == np.uint64
    assert np.array_equal(oracle.synth_uniform(2, 5, 3), oracle.synth_uniform(2, 0, 8)[5:8])
    si = oracle.synth_small_int(3, 0, 1000)
    assert set(np.unique(si)) == {-3.0, -2.0, -1.0, 0.0, 1.0, 2.0, 3.0}
    assert abs(float(oracle.synth_uniform(9, 0, 200000).mean())) < 0.01
    # splitmix64 reference value (x = 0 -> first output of the published generator)
    assert int(oracle._splitmix64(np.zeros(1, dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF


def test_topk_ties_and_nan():
    s = np.array([1.0, 3.0, 3.0, np.nan, -np.inf, 3.0, 2.0])
    v, i = oracle.topk_desc(s, 5)
    assert i.tolist() == [1, 2, 5, 6, 0]
    v, i = oracle.topk_desc(s, 7)
    assert i.tolist()[-2:] == [4, 3] and np.isnan(v[-1])


def test_num_hits_formula():
    # `_search.py:66-67`: defaults -> 4 * 10 = 40 rows; test_rerank uses num_results=40 -> 160
    assert oracle.num_hits(3) == 40 and oracle.num_hits(40) == 160
    assert oracle.num_hits(8, oversample=4, chunk_max_size=1024) == 2 * 10


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_two_stage_semantics(metric):
    rng = np.random.default_rng(3)
    E = rng.standard_normal((300, 16)).astype(np.float32)
    off = ragged_offsets(rng, 300, 1, 9)
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    q = rng.standard_normal(16).astype(np.float32)
    s, c = oracle.search_chunks(E, r2c, q, 40, 5, metric)
    sim = oracle.similarity(E, q, metric)
    rows = np.argsort(-sim, kind="stable")[:40]
    best = {}
    for r in rows:
        best[r2c[r]] = max(best.get(r2c[r], -np.inf), sim[r])
    exp = sorted(best.items(), key=lambda kv: (-kv[1], kv[0]))[:5]
    assert c.tolist() == [e[0] for e in exp]
    np.testing.assert_allclose(s, [e[1] for e in exp])
    # fewer chunks than requested when the hits span fewer chunks
    s2, c2 = oracle.search_chunks(E, np.zeros(300, dtype=int), q, 40, 5, metric)
    assert len(c2) == 1


def test_maxsim_reduces_to_single_vector_max():
    rng = np.random.default_rng(4)
    D = rng.standard_normal((200, 24))
    off = ragged_offsets(rng, 200, 1, 7, empty_every=5)
    q = rng.standard_normal(24)
    ms = oracle.maxsim_scores(D, off, q[None, :])
    dots = D @ q
    for c in range(len(off) - 1):
        b, e = off[c], off[c + 1]
        if e > b:
            assert ms[c] == pytest.approx(dots[b:e].max())
        else:
            assert np.isneginf(ms[c])
    cand = np.array([0, 3, 3, len(off) - 2])
    np.testing.assert_allclose(oracle.maxsim_candidates(D, off, q[None, :], cand), ms[cand])


def test_merge_equals_single_shard():
    rng = np.random.default_rng(5)
    E = rng.integers(-3, 4, size=(500, 32)).astype(np.float32)  # many ties
    q = rng.integers(-3, 4, size=32).astype(np.float32)
    full_s, full_i = oracle.search_rows(E, q, 20, "dot")
    parts_s, parts_i = [], []
    for lo, hi in [(0, 170), (170, 340), (340, 500)]:
        s, i = oracle.search_rows(E[lo:hi], q, 20, "dot")
        parts_s.append(s)
        parts_i.append(i + lo)
    ms, mi = oracle.merge_topk(parts_s, parts_i, 20)
    assert np.array_equal(mi, full_i) and np.array_equal(ms, full_s)


def test_shard_bounds():
    off = np.array([0, 3, 3, 10, 11, 20, 20, 31])
    for w in (1, 2, 3, 4, 8):
        b = oracle.shard_bounds_by_chunk(off, w)
        assert b[0][0] == 0 and b[-1][1] == len(off) - 1
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


# ---- a6 in DuckDB's own formulation (parity unpinned: DuckDB is absent; this MEASURES the gap between formulations) ----------------
DUCKDB_GAP_ULPS = 32  # |repository form - DuckDB form| in units of 2^-24 x max(1, |similarity| scale); measured: <= 24 (l2, unit rows)


@pytest.mark.parametrize("kind", ["unit_fp16", "uniform"])
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_duckdb_fp32_formulation_gap_is_bounded(kind, metric):
    """`/root/reference/src/raglite/_typing.py:123-134` evaluates the distance inside DuckDB: float32, element-order sums, ONE square
    root of the product of the squared norms for the cosine, a clamp to [-1, 1].  The repository's as-computed form (two roots and a
    product, blocked sums) differs from it by a few float32 ulps -- measured here, on fp16-rounded unit rows (what RAGLite stores,
    `_embed.py:138-140`) and on the benchmark's U(-1, 1) rows; both stay ~3 orders of magnitude inside the 1e-4 bar against float64."""
    n, d = 3000, 1024
    E = oracle.synth_matrix(11, n, d)
    q = oracle.synth_matrix(12, 1, d)[0]
    if kind == "unit_fp16":
        E = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
        q = (q / np.linalg.norm(q)).astype(np.float16).astype(np.float32)
    ours = (1.0 - oracle.distance(E, q, metric, np.float32)).astype(np.float64)
    duck = oracle.similarity_duckdb_fp32(E, q, metric).astype(np.float64)
    truth = 1.0 - oracle.distance(E, q, metric, np.float64)
    unit = 2.0 ** -24 * max(1.0, float(np.abs(truth).max()))
    assert float(np.abs(ours - duck).max()) <= DUCKDB_GAP_ULPS * unit
    assert float(np.abs(duck - truth).max()) <= DUCKDB_GAP_ULPS * unit  # the sequential float32 sums are the less accurate of the two
    assert float(np.abs(ours - truth).max()) <= 8 * unit
    if kind == "unit_fp16" or metric == "cosine":  # similarities of scale 1: the literal 1e-4 of north_star, with a factor 10 to spare
        assert DUCKDB_GAP_ULPS * unit < 1e-5


def test_duckdb_fp32_cosine_clamps_and_uses_one_root():
    """The two features of DuckDB's cosine this repository's form does not have: the similarity is clamped to [-1, 1] (so the distance
    of a vector to itself is >= 0, never -1e-7), and norms multiply BEFORE the root (a row of norm 1e-30 against a query of norm 1e30:
    the product is finite where either root alone is fine too -- and a row of norm 1e-25 squared underflows in both forms)."""
    e = np.full((1, 8), 0.35355338, dtype=np.float32)
    d = oracle.distance_duckdb_fp32(e, e[0], "cosine")
    assert d.dtype == np.float32 and 0.0 <= float(d[0]) <= 2.0 ** -22
    z = np.zeros((1, 8), dtype=np.float32)
    assert np.isnan(oracle.distance_duckdb_fp32(z, e[0], "cosine")[0])  # 0 / 0: NaN in DuckDB as well (ranks last here)
    assert float(oracle.distance_duckdb_fp32(e, e[0], "dot")[0]) == -float(np.float32(8 * np.float32(0.35355338) ** 2))
