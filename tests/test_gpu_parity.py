"""Parity of the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bars (BASELINE.json north_star): bit-exact indices on integer tie-free... in fact on ALL integer data
(ties resolve to the lowest id on both sides); cosine / MaxSim scores within 1e-4 in fp32; fp16
pooling outputs bit-exact against the reference's golden vectors.
"""

import json

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from oracle.fake_embedder import FakeLlama, make_sentences
from tests.util import assert_topk_close, ragged_offsets, sim_fp32_exact

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available()
    raglite_amd.set_device(0)
    return torch


# ---------------------------------------------------------------------------------------------------
# synthetic generator
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "small_int"])
def test_synth_bit_identical(torch_cuda, kind):
    t = torch_cuda.empty(100_003, dtype=torch_cuda.float32, device="cuda")
    raglite_amd.synth_fill(t, seed=17, start=12345, kind=kind)
    gen = oracle.synth_uniform if kind == "uniform" else oracle.synth_small_int
    assert np.array_equal(t.cpu().numpy().view(np.uint32), gen(17, 12345, 100_003).view(np.uint32))


# ---------------------------------------------------------------------------------------------------
# a1 + a2 + a3: pooling
# ---------------------------------------------------------------------------------------------------
def _golden_cases(kind):
    from pathlib import Path

    man = json.loads((Path(__file__).parent / "golden" / "manifest.json").read_text())
    return [(k, v) for k, v in man.items() if v["kind"] == kind]


@pytest.mark.parametrize("name,meta", _golden_cases("late_chunking"))
def test_embed_golden_late_chunking(golden_dir, name, meta):
    """`embed_strings()` of the mirror (HIP pooling) == the reference's own output, bit for bit."""
    golden = np.load(golden_dir / f"{name}.npz")["output"]
    emb = FakeLlama(dim=meta["dim"], n_ctx=meta["n_ctx"], n_batch=meta["n_batch"], seed=meta["embedder_seed"])
    cfg = raglite_amd.HotPathConfig(embedder=f"llama-cpp-python/fake/{name}", embedder_normalize=meta["normalize"])
    out = raglite_amd.embed_strings(make_sentences(meta["sentence_seed"], meta["n_sentences"]), config=cfg, embedder=emb)
    assert out.dtype == np.float16 and out.shape == golden.shape and np.all(np.isfinite(out))
    assert np.array_equal(out.view(np.uint16), golden.view(np.uint16))
    if meta["normalize"]:  # tests/test_embed.py:26 of the reference
        assert np.allclose(np.linalg.norm(out.astype(np.float64), axis=1), 1.0, rtol=1e-3)


@pytest.mark.parametrize("name,meta", _golden_cases("batch"))
def test_embed_golden_batch(golden_dir, name, meta):
    golden = np.load(golden_dir / f"{name}.npz")["output"]
    emb = FakeLlama(dim=meta["dim"], n_ctx=meta["n_ctx"], seed=meta["embedder_seed"])
    cfg = raglite_amd.HotPathConfig(embedder=f"llama-cpp-python/fake/{name}", embedder_normalize=meta["normalize"])
    out = raglite_amd.embed_strings_without_late_chunking(
        make_sentences(meta["sentence_seed"], meta["n_sentences"]), config=cfg, embedder=emb)
    assert np.array_equal(out.view(np.uint16), golden.view(np.uint16))


@pytest.mark.parametrize("dim", [4, 48, 64, 100, 257, 768, 1024, 2048, 4096])
@pytest.mark.parametrize("normalize,eps", [(True, 0.0), (True, 2.2e-16), (False, 0.0)])
def test_pool_norm_vs_oracle(dim, normalize, eps):
    rng = np.random.default_rng(dim)
    T = 600
    tokens = oracle.synth_matrix(dim, T, dim)
    cuts = np.sort(rng.choice(np.arange(1, T), size=60, replace=False))
    begins = np.concatenate(([0], cuts)).astype(np.int64)
    ends = np.concatenate((cuts, [T])).astype(np.int64)
    begins = np.concatenate((begins, [5, 10, 0]))  # overlapping spans, an EMPTY span, the whole matrix
    ends = np.concatenate((ends, [300, 10, T]))
    f32, f16 = raglite_amd.pool_norm(tokens, begins, ends, normalize=normalize, eps=eps, want_f32=True, want_f16=True)
    ref64, ref16 = oracle.pool_norm_cast(tokens, begins, ends, normalize=normalize, eps=eps or None)
    empty = ends == begins
    assert np.all(np.isnan(f32[empty])) and np.all(np.isnan(f16[empty]))  # np.mean of zero rows
    np.testing.assert_allclose(f32[~empty], ref64[~empty], rtol=0, atol=1e-6)
    ulp = np.abs(f16[~empty].view(np.int16).astype(np.int32) - ref16[~empty].view(np.int16).astype(np.int32))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3, "fp16 cast differs from astype(float16)"


def test_pool_norm_zero_vector_eps_guard():
    tokens = np.zeros((8, 64), dtype=np.float32)
    _, guarded = raglite_amd.pool_norm(tokens, np.array([0]), np.array([8]), normalize=True, eps=2.2e-16)
    assert np.all(guarded == 0)  # `_embed.py:160-163`: 0 / max(0, eps) = 0
    _, raw = raglite_amd.pool_norm(tokens, np.array([0]), np.array([8]), normalize=True, eps=0.0)
    assert np.all(np.isnan(raw))  # `_embed.py:139`: 0 / 0


def test_pool_norm_device_pointers(torch_cuda):
    tokens = oracle.synth_matrix(3, 300, 1024)
    b = np.arange(0, 300, 10, dtype=np.int64)
    e = b + 10
    _, host = raglite_amd.pool_norm(tokens, b, e)
    t = torch_cuda.as_tensor(tokens, device="cuda")
    _, dev = raglite_amd.pool_norm(t, torch_cuda.as_tensor(b, device="cuda"), torch_cuda.as_tensor(e, device="cuda"))
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy().view(np.uint16), host.view(np.uint16))


# ---------------------------------------------------------------------------------------------------
# a5: adapter
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim,B", [(64, 1), (1024, 1), (1024, 5), (768, 3), (100, 2), (1024, 40), (1024, 1000), (384, 97), (64, 130)])
def test_adapter_apply(dim, B):
    """Every batch class: VALU scan (B <= 4), MFMA stream kernel (5..95), fp32 MFMA GEMM (B >= 96; cfg 5 uses 1000)."""
    A = oracle.synth_matrix(21, dim, dim)
    Q = oracle.synth_matrix(22, B, dim)
    out = raglite_amd.adapter_apply(A, Q if B > 1 else Q[0])
    ref = oracle.adapter_apply(A.astype(np.float64), (Q if B > 1 else Q[0]).astype(np.float64))
    np.testing.assert_allclose(out, ref, rtol=0, atol=TOL)
    Ai, Qi = oracle.synth_matrix(23, dim, dim, "small_int"), oracle.synth_matrix(24, B, dim, "small_int")
    outi = raglite_amd.adapter_apply(Ai, Qi)
    assert np.array_equal(outi, (Qi.astype(np.float64) @ Ai.astype(np.float64).T).astype(np.float32))  # exact
    h = raglite_amd.adapter_apply(A, Q, want_f16=True)  # `.astype(q.dtype)` with an fp16 query, `_search.py:62`
    assert h.dtype == np.float16
    np.testing.assert_allclose(h.astype(np.float64), ref.reshape(B, dim), rtol=2e-3, atol=1e-3)


# ---------------------------------------------------------------------------------------------------
# a6 + a7: similarity + exact top-k
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("n,dim", [(3000, 1024), (1500, 128), (700, 100), (5000, 384)])
def test_search_rows_integer_data_bit_exact(metric, n, dim):
    """Integer-valued embeddings: scores and indices bit-identical to the fp32 as-computed oracle."""
    E = oracle.synth_matrix(31, n, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    for b in range(2):
        q = oracle.synth_matrix(32 + b, 1, dim, "small_int")[0]
        s, r = idx.search_rows(q, 100)
        ref = sim_fp32_exact(E, q, metric)
        es, ei = oracle.topk_desc(ref, 100)
        assert np.array_equal(r, ei), "indices differ (ties must resolve to the lowest row)"
        assert np.array_equal(s.view(np.uint32), es.astype(np.float32).view(np.uint32)), "scores not bit-exact"
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 7, 33])
def test_search_rows_uniform_data_batched(metric, B):
    """U(-1,1) data, every batch size class: VALU scan (B <= 4) and MFMA tile path (B > 4, dim 1024)."""
    n, dim = 4100, 1024
    E = oracle.synth_matrix(41, n, dim)
    Q = oracle.synth_matrix(42, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, 50)
    for b in range(B):
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], metric), 50, TOL)
    idx.close()


@pytest.mark.parametrize("kind", ["unit_fp16", "uniform"])
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("B", [1, 33, 130])
def test_search_rows_against_duckdb_fp32_formulation(metric, kind, B):
    """a6 / a7 against the as-computed variant in DuckDB's own formulation (`oracle.distance_duckdb_fp32`: float32, element-order
    sums, one square root of the product of the squared norms, clamp -- `/root/reference/src/raglite/_typing.py:123-134` calls DuckDB's
    `array_cosine_distance` / `array_negative_inner_product` / `array_distance`): every similarity of every row (k = n) through the
    single-query, the batched and the GEMM kernels stays within the measured gap between the two formulations
    (tests/test_oracle_props.py: DUCKDB_GAP_ULPS) + 2 ulp of the score scale."""
    from tests.test_oracle_props import DUCKDB_GAP_ULPS

    n, dim = 1800, 1024
    E = oracle.synth_matrix(8100, n, dim)
    Q = oracle.synth_matrix(8101, B, dim)
    if kind == "unit_fp16":
        E = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, n)
    for b in sorted({0, B // 2, B - 1}):
        duck = oracle.similarity_duckdb_fp32(E, Q[b], metric).astype(np.float64)
        got = np.empty(n, dtype=np.float64)
        got[R[b]] = S[b]
        unit = 2.0 ** -24 * max(1.0, float(np.abs(duck).max()))
        assert float(np.abs(got - duck).max()) <= (DUCKDB_GAP_ULPS + 2) * unit
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("n,dim,B", [(1000, 1024, 96), (4100, 1024, 130), (777, 128, 257), (2049, 384, 128), (130, 64, 100)])
def test_search_rows_gemm_path_integer_bit_exact(metric, n, dim, B):
    """B >= 96 takes the 128 x 128-tiled fp32 MFMA GEMM (score_gemm.hip): ragged tile edges in both directions,
    every fast-path dim class; integer data => scores and indices bit-identical to the fp32 as-computed oracle,
    and identical to what the single-query VALU path returns (position- and batch-independence)."""
    E = oracle.synth_matrix(51, n, dim, "small_int")
    Q = oracle.synth_matrix(52, B, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    k = 40
    S, R = idx.search_rows(Q, k)
    for b in (0, 1, 63, 64, B // 2, B - 2, B - 1):
        ref = sim_fp32_exact(E, Q[b], metric)
        es, ei = oracle.topk_desc(ref, k)
        assert np.array_equal(R[b], ei), f"query {b}: indices differ"
        assert np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32)), f"query {b}: scores"
        s1, r1 = idx.search_rows(Q[b], k)
        assert np.array_equal(r1, R[b]) and np.array_equal(s1, S[b])
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_search_rows_gemm_path_uniform_tolerance(metric):
    n, dim, B = 3001, 1024, 200
    E = oracle.synth_matrix(53, n, dim)
    Q = oracle.synth_matrix(54, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, 50)
    for b in (0, 17, 127, 128, 199):
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], metric), 50, TOL)
    S2, R2 = idx.search_rows(Q, 50)
    assert np.array_equal(S, S2) and np.array_equal(R, R2)  # deterministic
    # a row's score does not depend on where the row sits: shift the corpus by 37 rows
    idx2 = raglite_amd.DeviceIndex(np.ascontiguousarray(E[37:]), metric=metric)
    S3, R3 = idx2.search_rows(Q, 50)
    for b in (0, 199):
        keep = R[b] >= 37
        assert np.array_equal(R[b][keep] - 37, R3[b][: keep.sum()]) and np.array_equal(S[b][keep], S3[b][: keep.sum()])
    idx.close()
    idx2.close()


def test_search_rows_edge_cases(torch_cuda):
    dim = 64
    E = oracle.synth_matrix(51, 37, dim)
    q = oracle.synth_matrix(52, 1, dim)[0]
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    s, r = idx.search_rows(q, 100)  # k > n: padding
    assert_topk_close(s, r, oracle.similarity(E, q, "cosine"), 100, TOL)
    assert np.all(r[37:] == -1) and np.all(np.isneginf(s[37:]))
    s1, r1 = idx.search_rows(q, 1)
    assert r1[0] == r[0]
    with pytest.raises(ValueError):
        idx.search_rows(q, 4096)
    with pytest.raises(ValueError):
        idx.search_rows(np.zeros(dim + 1, np.float32), 5)
    idx.close()
    empty = raglite_amd.DeviceIndex(np.zeros((0, dim), np.float32))  # empty DB, tests/test_search.py:76-85
    s, r = empty.search_rows(q, 5)
    assert np.all(r == -1) and np.all(np.isneginf(s))
    empty.close()
    # identical rows (the reference's np.ones fixture, tests/test_split_chunks.py:28): N-way tie -> slow path
    ones = np.ones((6000, 768), dtype=np.float16).astype(np.float32)
    idx = raglite_amd.DeviceIndex(ones, metric="cosine")
    s, r = idx.search_rows(np.ones(768, np.float32), 300)
    assert r.tolist() == list(range(300)) and np.allclose(s, 1.0, atol=1e-6)
    idx.close()
    # device-pointer call path returns CUDA tensors with the same bits
    Ed = torch_cuda.as_tensor(E, device="cuda")
    idx = raglite_amd.DeviceIndex(Ed, metric="dot")
    sd, rd = idx.search_rows(torch_cuda.as_tensor(q, device="cuda"), 10)
    sh, rh = idx.search_rows(q, 10)
    assert sd.is_cuda and np.array_equal(sd.cpu().numpy(), sh) and np.array_equal(rd.cpu().numpy(), rh)
    idx.close()


def test_topk_kernel_properties():
    rng = np.random.default_rng(61)
    for n, k in [(1, 1), (63, 64), (5000, 100), (100_000, 2048), (300_000, 10)]:
        x = rng.standard_normal(n).astype(np.float32)
        s, i = raglite_amd.topk(x, k)
        es, ei = oracle.topk_desc(x, k)
        kk = min(n, k)
        assert np.array_equal(i[:kk], ei) and np.array_equal(s[:kk], es)
    # specials: NaN last, -inf before NaN, +inf first, signed zeros tie -> lowest index
    x = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0, np.nan, -1.0, 1.0], dtype=np.float32)
    s, i = raglite_amd.topk(x, 9)
    assert i.tolist()[:3] == [3, 5, 8] and i.tolist()[-3:] == [4, 2, 6]
    assert np.isnan(s[-1]) and np.isnan(s[-2]) and np.isneginf(s[-3])
    # heavy ties straddling the candidate capacity: few distinct values
    x = rng.integers(0, 3, size=50_000).astype(np.float32)
    s, i = raglite_amd.topk(x, 1000)
    es, ei = oracle.topk_desc(x, 1000)
    assert np.array_equal(i, ei)
    # batched
    X = rng.standard_normal((7, 9000)).astype(np.float32)
    S, I = raglite_amd.topk(X, 33)
    for b in range(7):
        es, ei = oracle.topk_desc(X[b], 33)
        assert np.array_equal(I[b], ei) and np.array_equal(S[b], es)


# ---------------------------------------------------------------------------------------------------
# a6 + a7 + a8: two-stage semantics
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_search_chunks_two_stage(metric):
    rng = np.random.default_rng(71)
    n, dim = 2500, 1024
    E = oracle.synth_matrix(72, n, dim, "small_int")
    off = ragged_offsets(rng, n, 1, 9)
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    idx = raglite_amd.DeviceIndex(E, off, metric=metric)
    for seed, (hits, k) in enumerate([(40, 3), (40, 10), (160, 40), (12, 12)]):
        q = oracle.synth_matrix(80 + seed, 1, dim, "small_int")[0]
        s, c, cnt = idx.search_chunks(q, hits, k)
        rs, rr = oracle.topk_desc(sim_fp32_exact(E, q, metric), hits)
        es, ec = oracle.group_chunk_max(rs, r2c[rr], k)
        assert cnt == len(ec) and c[:cnt].tolist() == ec.tolist()
        assert np.array_equal(s[:cnt], es.astype(np.float32))
        assert np.all(c[cnt:] == -1)
    # all hits inside one chunk -> a single result even though k = 5 (reference semantics)
    one = raglite_amd.DeviceIndex(E[:50], np.array([0, 50]), metric=metric)
    s, c, cnt = one.search_chunks(E[3], 40, 5)
    assert cnt == 1 and c[0] == 0
    one.close()
    idx.close()


# ---------------------------------------------------------------------------------------------------
# a9: MaxSim
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nq", [1, 7, 16, 17, 32])
@pytest.mark.parametrize("layout", ["fixed8", "ragged", "rows", "ragged_with_empty", "one_big"])
def test_maxsim_stream_d1024_integer_exact(nq, layout):
    """dim 1024 MFMA streaming kernel; integer data -> exact scores, exact top-k incl. tie order."""
    rng = np.random.default_rng(91)
    n, dim = 4133, 1024  # not a multiple of 16 nor of the grid
    E = oracle.synth_matrix(92, n, dim, "small_int")
    if layout == "fixed8":
        n = 4128
        E = E[:n]
        off = np.arange(0, n + 1, 8, dtype=np.int64)
    elif layout == "ragged":
        off = ragged_offsets(rng, n, 1, 15)
    elif layout == "rows":
        off = None
    elif layout == "ragged_with_empty":
        off = ragged_offsets(rng, n, 1, 40, empty_every=4)
    else:
        off = np.array([0, 5, n - 3, n], dtype=np.int64)  # one chunk spanning many workgroups' shares
    Q = oracle.synth_matrix(93, nq, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    off_ref = np.arange(n + 1) if off is None else off
    ref = oracle.maxsim_scores(E, off_ref, Q).astype(np.float32)
    got = idx.maxsim_scores(Q)
    assert np.array_equal(got, ref), f"max abs err {np.nanmax(np.abs(got - ref))}"
    s, c = idx.maxsim_topk(Q, 100)
    es, ec = oracle.topk_desc(ref, 100)
    kk = len(ec)
    assert np.array_equal(c[:kk], ec) and np.array_equal(s[:kk], es)
    idx.close()


def test_maxsim_stream_uniform_tolerance_and_linearity():
    n, dim, nq = 20_000, 1024, 32
    E = oracle.synth_matrix(94, n, dim)
    E /= np.linalg.norm(E, axis=1, keepdims=True)  # ColBERT convention: unit rows
    Q = oracle.synth_matrix(95, nq, dim)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    off = ragged_offsets(np.random.default_rng(96), n, 1, 15)
    idx = raglite_amd.DeviceIndex(E.astype(np.float32), off, metric="dot")
    got = idx.maxsim_scores(Q.astype(np.float32))
    ref = oracle.maxsim_scores(E.astype(np.float32), off, Q.astype(np.float32))
    np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)
    s, c = idx.maxsim_topk(Q.astype(np.float32), 100)
    assert_topk_close(s, c, ref, 100, TOL)
    # batched entry point.  Two queries share a pass of the streaming kernel: same bits as query-by-query calls.  Three or
    # more go through the eight-query kernel over the pre-split corpus image, which sums K in one chain instead of four
    # quarters: same bar against the oracle, not the same bits.
    Qb = np.stack([Q, Q[::-1].copy(), 0.5 * Q]).astype(np.float32)
    bs, bc = idx.maxsim_topk_batch(Qb[:2], 100)
    for b in range(2):
        ss, cc = idx.maxsim_topk(Qb[b], 100)
        assert np.array_equal(bs[b], ss) and np.array_equal(bc[b], cc)
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    for b in range(3):
        assert_topk_close(bs[b], bc[b], oracle.maxsim_scores(E.astype(np.float32), off, Qb[b]), 100, TOL)
    # scaling Q by a power of two scales every score exactly (size-independent property)
    got2 = idx.maxsim_scores((2.0 * Q).astype(np.float32))
    assert np.array_equal(got2, 2.0 * got)
    # nq = 1 reduces to the reference's single-vector per-chunk max (`_search.py:143-149`)
    one = idx.maxsim_scores(Q[:1].astype(np.float32))
    dots = E.astype(np.float64) @ Q[0].astype(np.float64)
    np.testing.assert_allclose(one, np.maximum.reduceat(dots, off[:-1]), rtol=0, atol=TOL)
    idx.close()


@pytest.mark.parametrize("dim", [128, 256, 384, 512, 768])
@pytest.mark.parametrize("nq", [3, 16, 32])
def test_maxsim_stream_other_dims(dim, nq):
    """The MFMA streaming kernel is templated on dim/4 columns per compute wave: every supported dim, both
    query-tile counts, ragged chunks with empties, exact on integer data; plus the batched row-score mode."""
    rng = np.random.default_rng(dim + nq)
    n = 3000 + dim // 64
    E = oracle.synth_matrix(130 + dim, n, dim, "small_int")
    off = ragged_offsets(rng, n, 1, 12, empty_every=9)
    Q = oracle.synth_matrix(131 + dim, nq, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    ref = oracle.maxsim_scores(E, off, Q).astype(np.float32)
    assert np.array_equal(idx.maxsim_scores(Q), ref)
    s, c = idx.maxsim_topk(Q, 50)
    es, ec = oracle.topk_desc(ref, 50)
    assert np.array_equal(c, ec) and np.array_equal(s, es)
    idx.close()
    cos = raglite_amd.DeviceIndex(E, metric="cosine")
    S, R = cos.search_rows(Q, 20)  # nq > 4 -> MFMA row-score mode + metric transform
    for b in range(0, nq, max(1, nq // 3)):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], "cosine"), 20)
        if nq > 4:
            assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], "cosine"), 20, TOL)
        else:
            assert np.array_equal(R[b], ei) and np.array_equal(S[b], es.astype(np.float32))
    cos.close()


@pytest.mark.parametrize("dim,nq", [(64, 5), (128, 40), (100, 3), (2048, 9)])
def test_maxsim_generic_path(dim, nq):
    rng = np.random.default_rng(dim)
    n = 900
    E = oracle.synth_matrix(97, n, dim, "small_int")
    off = ragged_offsets(rng, n, 1, 12, empty_every=7)
    Q = oracle.synth_matrix(98, nq, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    ref = oracle.maxsim_scores(E, off, Q).astype(np.float32)
    assert np.array_equal(idx.maxsim_scores(Q), ref)
    s, c = idx.maxsim_topk(Q, 20)
    es, ec = oracle.topk_desc(ref, 20)
    assert np.array_equal(c, ec) and np.array_equal(s, es)
    idx.close()


@pytest.mark.parametrize("dim,nq,rows", [(128, 32, 64), (128, 7, 64), (128, 20, "ragged"), (96, 6, "ragged"), (1024, 4, 8)])
def test_maxsim_rerank(dim, nq, rows):
    """SURVEY cfg 3: per-query candidate lists; dim 128 takes the MFMA fast path, others the generic one."""
    rng = np.random.default_rng(101)
    n_chunks, n_queries, n_cand = 300, 9, 37
    if rows == "ragged":
        off = ragged_offsets(rng, 4000, 1, 40, empty_every=11)
        n_chunks = len(off) - 1
    else:
        off = np.arange(0, n_chunks * rows + 1, rows, dtype=np.int64)
    n = int(off[-1])
    E = oracle.synth_matrix(102, n, dim, "small_int")
    Q = oracle.synth_matrix(103, n_queries * nq, dim, "small_int").reshape(n_queries, nq, dim)
    cand = rng.integers(0, n_chunks, size=(n_queries, n_cand)).astype(np.int32)
    cand[:, 1] = cand[:, 0]  # duplicates are legal
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    got = idx.maxsim_rerank(Q, cand)
    for i in range(n_queries):
        ref = oracle.maxsim_candidates(E, off, Q[i], cand[i]).astype(np.float32)
        assert np.array_equal(got[i], ref), f"query {i}"
    with pytest.raises(ValueError):
        idx.maxsim_rerank(Q, np.full((n_queries, 2), n_chunks, np.int32))
    # uniform data tolerance
    Eu = oracle.synth_matrix(104, n, dim) / np.sqrt(dim)
    Qu = oracle.synth_matrix(105, n_queries * nq, dim).reshape(n_queries, nq, dim) / np.sqrt(dim)
    iu = raglite_amd.DeviceIndex(Eu.astype(np.float32), off, metric="dot")
    gu = iu.maxsim_rerank(Qu.astype(np.float32), cand)
    for i in range(n_queries):
        np.testing.assert_allclose(gu[i], oracle.maxsim_candidates(Eu.astype(np.float32), off, Qu[i].astype(np.float32), cand[i]),
                                   rtol=0, atol=TOL)
    iu.close()
    idx.close()


# ---------------------------------------------------------------------------------------------------
# section 8e: merge
# ---------------------------------------------------------------------------------------------------
def test_merge_topk_kernel_and_shard_equivalence():
    n, dim, k = 6000, 1024, 100
    E = oracle.synth_matrix(111, n, dim, "small_int")
    Q = oracle.synth_matrix(112, 3, dim, "small_int")
    full = raglite_amd.DeviceIndex(E, metric="cosine")
    fs, fi = full.search_rows(Q, k)
    bounds = [(0, 1000), (1000, 1007), (1007, 4000), (4000, 6000)]
    ss, ii = [], []
    for lo, hi in bounds:
        sh = raglite_amd.DeviceIndex(E[lo:hi], metric="cosine")
        s, i = sh.search_rows(Q, k)
        ss.append(s)
        ii.append(np.where(i >= 0, i + lo, -1))
        sh.close()
    ms, mi = raglite_amd.merge_topk(np.stack(ss), np.stack(ii).astype(np.int32), k)
    assert np.array_equal(mi, fi) and np.array_equal(ms, fs)
    from raglite_amd import merge_topk_host

    hs, hi_ = merge_topk_host(np.stack(ss), np.stack(ii), k)
    assert np.array_equal(hi_, fi) and np.array_equal(hs, fs)
    full.close()


# ---------------------------------------------------------------------------------------------------
# end-to-end drop-ins
# ---------------------------------------------------------------------------------------------------
def test_vector_search_and_rerank_dropins():
    rng = np.random.default_rng(121)
    dim, n_chunks = 128, 60
    mats = [oracle.synth_matrix(122 + i, int(rng.integers(1, 7)), dim) for i in range(n_chunks)]
    mats = [(m / np.linalg.norm(m, axis=1, keepdims=True)).astype(np.float16) for m in mats]  # as stored by RAGLite
    ids = [f"{i:016x}" for i in range(n_chunks)]
    docs = [f"chunk body {i}" for i in range(n_chunks)]
    A = np.linalg.qr(rng.standard_normal((dim, dim)))[0]  # orthogonal adapter, `_query_adapter.py:202-205`
    meta = [{"topic": ["Physics"] if i % 2 == 0 else ["Biology"], "author": "Albert"} for i in range(n_chunks)]
    gi = raglite_amd.GpuIndex(ids, mats, query_adapter=A, docs=docs, metadata=meta)
    cfg = raglite_amd.HotPathConfig()
    q = mats[17][0]
    got_ids, got_scores = raglite_amd.vector_search(q, num_results=5, config=cfg, index=gi)
    assert len(got_ids) == len(got_scores) == 5 and all(isinstance(s, float) for s in got_scores)
    E = np.vstack(mats).astype(np.float32)
    r2c = np.repeat(np.arange(n_chunks), [len(m) for m in mats])
    qa = oracle.adapter_apply(A, q)  # fp64 matvec, cast back to fp16 (`_search.py:62`)
    es, ec = oracle.search_chunks(E, r2c, qa.astype(np.float32), oracle.num_hits(5), 5, "cosine")
    assert got_ids == [ids[c] for c in ec]
    np.testing.assert_allclose(got_scores, es, rtol=0, atol=2e-3)  # adapter output rounds through fp16
    cfg_na = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    ids_na, sc_na = raglite_amd.vector_search(q, num_results=3, config=cfg_na, index=gi)
    assert ids_na[0] == ids[17] and abs(sc_na[0] - 1.0) < 1e-3  # the query is a stored row
    # metadata filter (filter-first semantics, `_search.py:105-119`)
    f_ids, _ = raglite_amd.vector_search(q, num_results=4, metadata_filter={"topic": "Biology"}, config=cfg_na, index=gi)
    assert len(f_ids) == 4 and all(int(i, 16) % 2 == 1 for i in f_ids)
    assert raglite_amd.vector_search(q, num_results=4, metadata_filter={"topic": "Chemistry"}, config=cfg_na, index=gi) == ([], [])
    # reranker plugin: Kendall-tau style contract of tests/test_rerank.py:64-70 (best-first ordering)
    qv = np.vstack([mats[5].astype(np.float32), mats[9].astype(np.float32)])
    ranker = raglite_amd.MaxSimRanker(gi, lambda _q: qv)
    cfg_r = raglite_amd.HotPathConfig(reranker=ranker)
    class _Chunk:  # the reference passes Chunk objects whose str() is the chunk text (`_search.py:395`)
        def __init__(self, text): self.text = text
        def __str__(self): return self.text

    cand = [_Chunk(docs[i]) for i in (30, 9, 2, 5, 41)]
    out = raglite_amd.rerank_chunks("q", cand, config=cfg_r)
    ref = oracle.maxsim_candidates(E, np.concatenate(([0], np.cumsum([len(m) for m in mats]))), qv, [30, 9, 2, 5, 41])
    assert [c.text for c in out] == [cand[i].text for i in np.argsort(-ref, kind="stable")]
    assert {c.text for c in out[:2]} == {docs[5], docs[9]}
    # chunk ids resolve through chunk_lookup (the reference's retrieve_chunks)
    by_id = {ids[i]: _Chunk(docs[i]) for i in range(n_chunks)}
    out_ids = raglite_amd.rerank_chunks("q", [ids[i] for i in (30, 9, 2, 5, 41)], config=cfg_r,
                                        chunk_lookup=lambda cids: [by_id[c] for c in cids])
    assert [c.text for c in out_ids] == [c.text for c in out]
    # lifecycle through the host mirror: inserted chunks are found, deleted chunks never come back
    extra = [(m / np.linalg.norm(m, axis=1, keepdims=True)).astype(np.float16)
             for m in (oracle.synth_matrix(300 + i, 2, dim) for i in range(3))]
    gi.insert_chunks([f"new{i}" for i in range(3)], extra, docs=[f"new body {i}" for i in range(3)],
                     metadata=[{"topic": ["Chemistry"]}] * 3)
    top, sc = raglite_amd.vector_search(extra[1][0], num_results=3, config=cfg_na, index=gi)
    assert top[0] == "new1" and abs(sc[0] - 1.0) < 1e-3
    chem, _ = raglite_amd.vector_search(q, num_results=5, metadata_filter={"topic": "Chemistry"}, config=cfg_na, index=gi)
    assert sorted(chem) == ["new0", "new1", "new2"]
    assert gi.delete_chunks(["new1", ids[17]]) == 2
    top2, _ = raglite_amd.vector_search(extra[1][0], num_results=8, config=cfg_na, index=gi)
    assert "new1" not in top2 and len(top2) == 8
    top3, _ = raglite_amd.vector_search(q, num_results=3, config=cfg_na, index=gi)
    assert ids[17] not in top3
    gi.close()


# ---------------------------------------------------------------------------------------------------
# 8f-1: metadata filter pushed down, deleted chunks, appended rows
# ---------------------------------------------------------------------------------------------------
def _r2c(off):
    return np.repeat(np.arange(len(off) - 1), np.diff(off))


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("density", [0.5, 0.02, 0.0])
def test_filtered_search_matches_oracle(metric, density):
    """Filter-first branch (`_search.py:105-119`) as a device bitset: integer data => bit-exact against the oracle,
    including filters that leave fewer rows than num_hits (padding) and filters that match nothing."""
    rng = np.random.default_rng(7)
    n, dim = 6000, 256
    off = ragged_offsets(rng, n, 1, 9)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(61, n, dim, "small_int")
    q = oracle.synth_matrix(62, 2, dim, "small_int")
    ok = rng.random(n_chunks) < density
    idx = raglite_amd.DeviceIndex(E, off, metric=metric)
    r2c = _r2c(off)
    for b in range(2):
        ref = sim_fp32_exact(E, q[b], metric)
        es, er = oracle.topk_desc(np.where(ok[r2c], ref, -np.inf), 80)
        dead = (er >= 0) & ~ok[r2c][np.clip(er, 0, n - 1)]
        es, er = np.where(dead, -np.inf, es), np.where(dead, -1, er)
        s, r = idx.search_rows(q[b], 80, chunk_filter=ok)
        assert np.array_equal(r, er) and np.array_equal(s, es.astype(np.float32))
        # two-stage semantics with the filter
        ws, wc = oracle.search_chunks_filtered(E, r2c, q[b], 80, 10, ok, metric)
        gs, gc, cnt = idx.search_chunks(q[b], 80, 10, chunk_filter=ok)
        assert int(cnt) == len(wc) and np.array_equal(gc[: len(wc)], wc)
        np.testing.assert_allclose(gs[: len(wc)], ws, atol=TOL)
    # MaxSim with the filter
    Q = oracle.synth_matrix(63, 7, dim, "small_int")
    ws, wc = oracle.maxsim_topk_filtered(E, off, Q, 50, ok)
    gs, gc = idx.maxsim_topk(Q, 50, chunk_filter=ok)
    assert np.array_equal(gc, wc) and np.array_equal(gs, ws.astype(np.float32))
    idx.close()


def test_filter_device_pointers_and_unfiltered_equivalence(torch_cuda):
    torch = torch_cuda
    n, dim = 5000, 1024
    rng = np.random.default_rng(8)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=64)
    q = torch.empty((3, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q, seed=65)
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    all_ok = np.ones(len(off) - 1, bool)
    s0, r0 = idx.search_rows(q, 64)
    s1, r1 = idx.search_rows(q, 64, chunk_filter=all_ok)
    assert torch.equal(s0, s1) and torch.equal(r0, r1)
    ok = rng.random(len(off) - 1) < 0.3
    s2, r2 = idx.search_rows(q, 64, chunk_filter=torch.as_tensor(ok, device="cuda"))
    r2c = _r2c(off)
    assert ok[r2c[r2.cpu().numpy()]].all()
    Eh = E.cpu().numpy()
    for b in range(3):
        assert_topk_close(s2[b].cpu().numpy(), r2[b].cpu().numpy(),
                          np.where(ok[r2c], oracle.similarity(Eh, q[b].cpu().numpy(), "cosine"), -np.inf), 64, TOL)
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_index_append_and_delete_lifecycle(metric):
    """insert_documents / delete_documents on the device image: an index grown by appends and thinned by deletes
    answers exactly like a fresh index over the surviving rows (ordinals mapped back), bit for bit."""
    rng = np.random.default_rng(9)
    dim = 384
    parts = [(1500, 61), (700, 62), (1, 63), (2200, 64)]
    Es, offs = [], []
    for n, seed in parts:
        Es.append(oracle.synth_matrix(seed, n, dim, "small_int"))
        offs.append(ragged_offsets(rng, n, 1, 7))
    idx = raglite_amd.DeviceIndex(Es[0], offs[0], metric=metric)
    for E, off in zip(Es[1:], offs[1:]):
        idx.append(E, np.diff(off))
    E_all = np.concatenate(Es)
    off_all = np.concatenate([[0]] + [o[1:] + sum(len(e) for e in Es[:i]) for i, o in enumerate(offs)]).astype(np.int64)
    assert idx.n_rows == len(E_all) and idx.n_chunks == len(off_all) - 1
    assert np.array_equal(idx.chunk_offsets, off_all)
    fresh = raglite_amd.DeviceIndex(E_all, off_all, metric=metric)
    q = oracle.synth_matrix(66, 3, dim, "small_int")
    Q = oracle.synth_matrix(67, 5, dim, "small_int")
    for b in range(3):
        a, f = idx.search_rows(q[b], 100), fresh.search_rows(q[b], 100)
        assert np.array_equal(a[0], f[0]) and np.array_equal(a[1], f[1])
    a, f = idx.maxsim_topk(Q, 60), fresh.maxsim_topk(Q, 60)
    assert np.array_equal(a[0], f[0]) and np.array_equal(a[1], f[1])
    # delete a third of the chunks (some twice), including the very first and the very last
    n_chunks = len(off_all) - 1
    dead = np.unique(np.concatenate([rng.choice(n_chunks, n_chunks // 3, replace=False), [0, n_chunks - 1]]))
    idx.delete_chunks(dead)
    idx.delete_chunks(dead[:5])
    alive = np.ones(n_chunks, bool)
    alive[dead] = False
    r2c = _r2c(off_all)
    assert idx.live() == (int(alive[r2c].sum()), int(alive.sum()))
    for b in range(3):
        ws, wr = oracle.search_rows_filtered(E_all, r2c, q[b], 100, alive, metric, np.float64)
        s, r = idx.search_rows(q[b], 100)
        assert np.array_equal(r, wr)
        ref = sim_fp32_exact(E_all, q[b], metric)
        assert np.array_equal(s, ref[wr].astype(np.float32))
        gs, gc, cnt = idx.search_chunks(q[b], 40, 8)
        os_, oc = oracle.search_chunks_filtered(E_all, r2c, q[b], 40, 8, alive, metric)
        assert int(cnt) == len(oc) and np.array_equal(gc[: len(oc)], oc)
    ws, wc = oracle.maxsim_topk_filtered(E_all, off_all, Q, 60, alive)
    gs, gc = idx.maxsim_topk(Q, 60)
    assert np.array_equal(gc, wc) and np.array_equal(gs, ws.astype(np.float32))
    sb, cb = idx.maxsim_topk_batch(np.stack([Q, Q]), 60)
    assert np.array_equal(cb[0], wc) and np.array_equal(cb[1], wc)
    assert np.isneginf(idx.maxsim_scores(Q)[dead]).all()
    # deletes combine with a filter, and appends after deletes are live
    flt = rng.random(n_chunks) < 0.5
    ws, wr = oracle.search_rows_filtered(E_all, r2c, q[0], 50, alive & flt, metric)
    s, r = idx.search_rows(q[0], 50, chunk_filter=flt)
    assert np.array_equal(r, wr)
    extra = oracle.synth_matrix(68, 40, dim, "small_int")
    idx.append(extra)  # one chunk per row
    E2 = np.concatenate([E_all, extra])
    off2 = np.concatenate([off_all, off_all[-1] + 1 + np.arange(40)])
    alive2 = np.concatenate([alive, np.ones(40, bool)])
    ws, wr = oracle.search_rows_filtered(E2, _r2c(off2), q[1], 100, alive2, metric)
    s, r = idx.search_rows(q[1], 100)
    assert np.array_equal(r, wr)
    idx.close()
    fresh.close()


# ---------------------------------------------------------------------------------------------------
# 8f-1: fp16-stored corpus (the reference's own storage precision, `_embed.py:140`)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
@pytest.mark.parametrize("n,dim", [(3000, 1024), (1501, 128), (2050, 384), (777, 768)])
def test_f16_storage_search_rows_integer_bit_exact(metric, n, dim):
    """Integer data is exact in fp16: every batch class of an fp16-stored index (VALU scan16 for B <= 4, f16-MFMA
    stream passes beyond) returns the oracle's scores and indices bit for bit -- and so equals the fp32-stored index."""
    E = oracle.synth_matrix(71, n, dim, "small_int")
    idx16 = raglite_amd.DeviceIndex(E, metric=metric, storage="f16")
    idx32 = raglite_amd.DeviceIndex(E, metric=metric)
    for B in (1, 3, 7, 33):
        Q = oracle.synth_matrix(72 + B, B, dim, "small_int")
        S, R = idx16.search_rows(Q if B > 1 else Q[0], 60)
        S, R = np.atleast_2d(S), np.atleast_2d(R)
        S32, R32 = idx32.search_rows(Q if B > 1 else Q[0], 60)
        assert np.array_equal(S, np.atleast_2d(S32)) and np.array_equal(R, np.atleast_2d(R32))
        for b in (0, B - 1):
            es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), 60)
            assert np.array_equal(R[b], ei) and np.array_equal(S[b], es.astype(np.float32))
    idx16.close()
    idx32.close()


@pytest.mark.parametrize("dim,nq", [(1024, 32), (1024, 5), (128, 17), (512, 16), (768, 1)])
def test_f16_storage_maxsim(dim, nq):
    rng = np.random.default_rng(10)
    n = 5003
    off = ragged_offsets(rng, n, 1, 15)
    # integer data: exact
    Ei = oracle.synth_matrix(81, n, dim, "small_int")
    Qi = oracle.synth_matrix(82, nq, dim, "small_int")
    idx = raglite_amd.DeviceIndex(Ei, off, metric="dot", storage="f16")
    ref = oracle.maxsim_scores(Ei, off, Qi)
    assert np.array_equal(idx.maxsim_scores(Qi), ref.astype(np.float32))
    s, c = idx.maxsim_topk(Qi, 50)
    es, ec = oracle.topk_desc(ref.astype(np.float32), 50)
    assert np.array_equal(c, ec) and np.array_equal(s, es)
    idx.close()
    # unit-norm rows rounded through fp16 (what RAGLite stores); fp32 queries that are NOT fp16-valued (hi + lo path)
    # and fp16-valued queries (lo == 0 path): within 1e-4 of the fp64 MaxSim over the stored values
    E = oracle.synth_matrix(83, n, dim)
    E16 = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16)
    Q = oracle.synth_matrix(84, nq, dim)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    idx = raglite_amd.DeviceIndex(E16, off, metric="dot", storage="f16")
    for qq in (Q, Q.astype(np.float16).astype(np.float32)):
        ref = oracle.maxsim_scores(E16.astype(np.float64), off, qq.astype(np.float64))
        got = idx.maxsim_scores(qq)
        np.testing.assert_allclose(got, ref, rtol=0, atol=TOL)
        s, c = idx.maxsim_topk(qq, 40)
        assert_topk_close(s, c, ref, 40, TOL)
    assert np.array_equal(idx.maxsim_scores(Q), idx.maxsim_scores(Q))  # deterministic
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "l2"])
def test_f16_storage_uniform_tolerance_and_lifecycle(metric):
    n, dim = 4100, 1024
    rng = np.random.default_rng(11)
    off = ragged_offsets(rng, n, 1, 9)
    E16 = oracle.synth_matrix(85, n, dim).astype(np.float16)
    Ev = E16.astype(np.float32)  # the stored values
    idx = raglite_amd.DeviceIndex(E16, off, metric=metric, storage="f16")
    for B in (1, 4, 9):
        Q = oracle.synth_matrix(86 + B, B, dim)
        S, R = idx.search_rows(Q, 50)
        for b in range(B):
            assert_topk_close(S[b], R[b], oracle.similarity(Ev, Q[b], metric), 50, TOL)
    # filter + delete + append (fp32 rows are rounded into the fp16 store)
    n_chunks = len(off) - 1
    r2c = _r2c(off)
    q = oracle.synth_matrix(90, 1, dim)[0]
    flt = rng.random(n_chunks) < 0.4
    s, r = idx.search_rows(q, 30, chunk_filter=flt)
    assert flt[r2c[r]].all()
    assert_topk_close(s, r, np.where(flt[r2c], oracle.similarity(Ev, q, metric), -np.inf), 30, TOL)
    idx.delete_chunks(np.nonzero(~flt)[0])
    s2, r2 = idx.search_rows(q, 30)
    assert np.array_equal(r2, r) and np.array_equal(s2, s)
    extra = oracle.synth_matrix(91, 33, dim)
    idx.append(extra)
    Ev2 = np.concatenate([Ev, extra.astype(np.float16).astype(np.float32)])
    alive = np.concatenate([flt[r2c], np.ones(33, bool)])
    s3, r3 = idx.search_rows(extra[7], 5)
    assert r3[0] == n + 7
    assert_topk_close(s3, r3, np.where(alive, oracle.similarity(Ev2, extra[7], metric), -np.inf), 5, TOL)
    # the rerank beyond dim 128 on an fp16-stored index (round 6): the pairs kernels over the stored halves
    Qr = oracle.synth_matrix(92, 6, dim).reshape(1, 6, dim)
    cand = np.array([[0, n_chunks - 1, 3]], np.int32)
    Ev2c = np.concatenate([Ev, extra.astype(np.float16).astype(np.float32)])
    off2 = np.concatenate((off, off[-1] + 1 + np.arange(33)))  # (appended rows: one chunk each)
    got = idx.maxsim_rerank(Qr, cand)
    want = oracle.maxsim_scores(Ev2c, off2, Qr[0], np.float64)[cand[0]]
    live = np.concatenate([flt, np.ones(33, bool)])[cand[0]]
    assert np.isneginf(got[0][~live]).all()
    np.testing.assert_allclose(got[0][live], want[live], rtol=0, atol=2e-6 * max(1.0, float(np.abs(want).max())))
    with pytest.raises(Exception):  # ... up to 32 query vectors
        idx.maxsim_rerank(np.zeros((1, 33, dim), np.float32), np.zeros((1, 2), np.int32))
    idx.close()
    with pytest.raises(Exception):
        raglite_amd.DeviceIndex(np.zeros((4, 100), np.float16), storage="f16")  # dim outside the fast path


@pytest.mark.parametrize("nq", [32, 9])
def test_f16_storage_rerank(nq):
    """cfg 3's shape class on an fp16-stored index: integer data exact, fp16 unit rows within 1e-4."""
    rng = np.random.default_rng(12)
    dim, n_queries, n_cand = 128, 5, 37
    off = ragged_offsets(rng, 6000, 1, 40)
    n, n_chunks = int(off[-1]), len(off) - 1
    cand = rng.integers(0, n_chunks, size=(n_queries, n_cand)).astype(np.int32)
    Ei = oracle.synth_matrix(95, n, dim, "small_int")
    Qi = oracle.synth_matrix(96, n_queries * nq, dim, "small_int").reshape(n_queries, nq, dim)
    idx = raglite_amd.DeviceIndex(Ei, off, metric="dot", storage="f16")
    got = idx.maxsim_rerank(Qi, cand)
    for b in range(n_queries):
        assert np.array_equal(got[b], oracle.maxsim_candidates(Ei, off, Qi[b], cand[b]).astype(np.float32))
    idx.close()
    E = oracle.synth_matrix(97, n, dim)
    E16 = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16)
    Q = oracle.synth_matrix(98, n_queries * nq, dim).reshape(n_queries, nq, dim)
    Q /= np.linalg.norm(Q, axis=2, keepdims=True)
    idx = raglite_amd.DeviceIndex(E16, off, metric="dot", storage="f16")
    got = idx.maxsim_rerank(Q, cand)
    for b in range(n_queries):
        np.testing.assert_allclose(got[b], oracle.maxsim_candidates(E16.astype(np.float64), off, Q[b], cand[b]),
                                   rtol=0, atol=TOL)
    idx.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("nq", [33, 64, 70])
def test_maxsim_more_than_32_query_vectors(storage, nq):
    """Passes of 32 query vectors accumulate into the chunk scores (fixed pass order): exact on integer data, and
    equal to the sum of the separately computed 32-vector blocks."""
    rng = np.random.default_rng(13)
    n, dim = 4001, 1024
    off = ragged_offsets(rng, n, 1, 15, empty_every=37)
    E = oracle.synth_matrix(101, n, dim, "small_int")
    Q = oracle.synth_matrix(102, nq, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
    got = idx.maxsim_scores(Q)
    assert np.array_equal(got, oracle.maxsim_scores(E, off, Q).astype(np.float32))
    s, c = idx.maxsim_topk(Q, 30)
    es, ec = oracle.topk_desc(oracle.maxsim_scores(E, off, Q).astype(np.float32), 30)
    assert np.array_equal(c, ec) and np.array_equal(s, es)
    Qu = oracle.synth_matrix(103, nq, dim)
    blocks = [idx.maxsim_scores(Qu[i : i + 32]).astype(np.float32) for i in range(0, nq, 32)]
    want = blocks[0]
    for b in blocks[1:]:
        want = want + b  # fp32, pass order
    assert np.array_equal(idx.maxsim_scores(Qu), want)
    idx.close()


def test_cfg1_shape_10k_chunks_top10():
    """BASELINE configs[0]: 10 k chunk embeddings, d = 1024, cosine top-10 -- the reference's CPU-runnable case, rows
    L2-normalised and rounded through fp16 like `_embed.py:139-140`; through the two-stage semantics with 5 rows/chunk."""
    n, dim = 10_000, 1024
    E = oracle.synth_matrix(1, n, dim)
    E = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    off = np.arange(0, n + 1, 5, dtype=np.int64)
    r2c = np.repeat(np.arange(n // 5), 5)
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    for seed in (2, 3):
        q = oracle.synth_matrix(seed, 1, dim)[0]
        s, r = idx.search_rows(q, 10)
        assert_topk_close(s, r, oracle.similarity(E, q, "cosine"), 10, TOL)
        gs, gc, cnt = idx.search_chunks(q, oracle.num_hits(10), 10)
        ws, wc = oracle.search_chunks(E, r2c, q, oracle.num_hits(10), 10, "cosine")
        assert int(cnt) == len(wc) and np.array_equal(gc[: len(wc)], wc)
        np.testing.assert_allclose(gs[: len(wc)], ws, rtol=0, atol=TOL)
    idx.close()


def test_torch_embedder_feeds_pooling_on_device(torch_cuda):
    """tokenise -> PyTorch-ROCm encoder -> `rl_pool_norm`, the token matrix never leaving HBM.  The
    pooled fp16 rows must equal the oracle's late-chunking pool of the very token matrices the encoder produced."""
    from raglite_amd import _embed
    from raglite_amd._torch_embedder import EncoderShape, TorchTokenEmbedder

    shape = EncoderShape(vocab_size=30000, hidden=256, layers=3, heads=8, ffn=512, max_positions=600, n_ctx=512)
    emb = TorchTokenEmbedder(shape, device="cuda", seed=5, n_batch=512)
    cfg = raglite_amd.HotPathConfig()
    sents = make_sentences(77, 60)  # ~1000 tokens: several segments with preambles
    out = raglite_amd.embed_strings(sents, config=cfg, embedder=emb)
    assert out.dtype == np.float16 and out.shape == (60, 256) and np.isfinite(out.astype(np.float32)).all()
    np.testing.assert_allclose(np.linalg.norm(out.astype(np.float64), axis=1), 1.0, rtol=1e-3)  # tests/test_embed.py:24-26
    tokens, begins, ends = _embed.plan_document(sents, emb)
    assert tokens.is_cuda
    _, want = oracle.pool_norm_cast(tokens.cpu().numpy().astype(np.float64), begins, ends, normalize=True, eps=None)
    assert np.array_equal(out.view(np.uint16), want.view(np.uint16))
    # the whole-string path (`_embed.py:144-165`)
    out2 = raglite_amd.embed_strings_without_late_chunking(sents[:7], config=cfg, embedder=emb)
    mats = [m.cpu().numpy() for m in emb.embed(sents[:7])]
    want2 = oracle.embed_string_batch_pool(mats, normalize=True)
    assert np.array_equal(out2.view(np.uint16), want2.view(np.uint16))


# ---------------------------------------------------------------------------------------------------
# 8f-3: device half of update_query_adapter
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_chunk_best_rows_and_gather(storage):
    rng = np.random.default_rng(14)
    n, dim = 3000, 256
    off = ragged_offsets(rng, n, 1, 9, empty_every=41)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(111, n, dim, "small_int")
    Q = oracle.synth_matrix(112, 6, dim, "small_int")
    cand = rng.integers(0, n_chunks, size=(6, 25)).astype(np.int32)
    cand[2, 3] = -1
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine", storage=storage)
    got = idx.chunk_best_rows(Q, cand)
    for b in range(6):
        for j, c in enumerate(cand[b]):
            if c < 0 or off[c + 1] == off[c]:
                assert got[b, j] == -1
            else:
                assert got[b, j] == off[c] + oracle.best_row(E[off[c] : off[c + 1]], Q[b])  # integer data: exact, ties -> first
    rows = np.asarray([0, 17, n - 1, 5], dtype=np.int32)
    assert np.array_equal(idx.gather_rows(rows), E[rows])
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_update_query_adapter_on_device(metric):
    """The batched fit (one rl_search_chunks over all evals -- the GEMM path at >= 96 evals --, one rl_chunk_best_rows,
    one rl_gather_rows, host NNLS + Procrustes) equals the per-eval loop of the oracle, and the fitted adapter is then
    used by vector_search."""
    rng = np.random.default_rng(15)
    dim, n_chunks = 128, 200
    mats = [oracle.synth_matrix(400 + i, int(rng.integers(1, 6)), dim) for i in range(n_chunks)]
    mats = [(m / np.linalg.norm(m, axis=1, keepdims=True)).astype(np.float16) for m in mats]
    ids = [f"{i:016x}" for i in range(n_chunks)]
    gi = raglite_amd.GpuIndex(ids, mats, metric=metric)
    E = np.vstack(mats).astype(np.float32)
    off = np.concatenate(([0], np.cumsum([len(m) for m in mats]))).astype(np.int64)
    evals = []
    for _ in range(120):
        t = int(rng.integers(0, n_chunks))
        q = (mats[t][0].astype(np.float32) + 0.08 * rng.standard_normal(dim)).astype(np.float16)
        evals.append((q, [ids[t], ids[(t + 11) % n_chunks]]))
    cfg = raglite_amd.HotPathConfig(vector_search_distance_metric=metric)
    A = raglite_amd.update_query_adapter(evals, optimize_top_k=10, config=cfg, index=gi)
    want, Qs, _ = oracle.update_query_adapter(evals, E, off, ids, optimize_top_k=10, metric=metric, dtype=np.float32)
    assert len(Qs) > 50
    np.testing.assert_allclose(A, want, rtol=0, atol=1e-6)
    assert gi.query_adapter is not None and gi.query_adapter.shape == (dim, dim)
    got_ids, _ = raglite_amd.vector_search(evals[0][0], num_results=5, config=cfg, index=gi)  # adapter now applied
    assert len(got_ids) == 5
    gi.close()


# ---------------------------------------------------------------------------------------------------
# 8f-4: semantic-chunking similarities on the device
# ---------------------------------------------------------------------------------------------------
def test_split_chunks_matches_reference_golden(torch_cuda):
    """`rl_partition_similarity` + host MILP vs the REAL reference `split_chunks` (tests/golden/split_chunks.npz):
    same cost vector within fp32 tolerance, identical chunks; batched documents == one at a time; CUDA tensors too."""
    from tests.test_oracle_golden import _split_cases

    torch = torch_cuda
    cases = _split_cases()
    for chunklets, X, max_size, cost, sizes, chunks in cases:
        got_chunks, got_embs = raglite_amd.split_chunks(chunklets, X, max_size=max_size)
        assert got_chunks == chunks and [len(m) for m in got_embs] == sizes.tolist()
        assert np.array_equal(np.vstack(got_embs), X)
        if len(cost):
            np.testing.assert_allclose(raglite_amd.partition_cost(chunklets, X), cost, rtol=0, atol=2e-6)
            t_chunks, t_embs = raglite_amd.split_chunks(chunklets, torch.as_tensor(X, device="cuda"), max_size=max_size)
            assert t_chunks == chunks and t_embs[0].is_cuda
    # all documents of equal dim in one launch
    same = [c for c in cases if c[1].shape[1] == 256 or c[1].shape[1] == 128]
    for dim in (128, 256):
        docs = [c for c in cases if c[1].shape[1] == dim]
        if not docs:
            continue
        docs = docs * 3
        Xall = np.vstack([c[1] for c in docs]).astype(np.float32)
        off = np.concatenate(([0], np.cumsum([len(c[1]) for c in docs])))
        lens = np.concatenate([[len(s) for s in c[0]] for c in docs])
        sim = raglite_amd.partition_similarities(Xall, off, lens)
        for d, c in enumerate(docs):
            one = raglite_amd.partition_similarities(c[1].astype(np.float32), np.asarray([0, len(c[1])]), [len(s) for s in c[0]])
            assert np.array_equal(sim[off[d] : off[d + 1]], one)
            assert sim[off[d + 1] - 1] == 0.0
    with pytest.raises(ValueError):
        raglite_amd.split_chunks(["x" * 50, "y"], np.ones((2, 8), np.float16), max_size=10)
    with pytest.raises(ValueError):
        raglite_amd.split_chunks(["x" * 9, "y" * 9], np.zeros((2, 8), np.float16), max_size=10)
    assert raglite_amd.split_chunks([], np.zeros((0, 8), np.float16))[0] == []


def test_end_to_end_indexing_pipeline_on_device(torch_cuda):
    """sentences -> embed_strings (PyTorch-ROCm encoder + rl_pool_norm) -> split_chunks (rl_partition_similarity + host
    MILP) -> GpuIndex.insert_chunks (rl_index_append) -> vector_search / MaxSim rerank: the reference's insert and
    search flow (`_insert.py:96-123`, `_search.py`) with every embedding-sized array produced and consumed in HBM."""
    from raglite_amd._torch_embedder import EncoderShape, TorchTokenEmbedder

    shape = EncoderShape(vocab_size=30000, hidden=128, layers=2, heads=4, ffn=256, max_positions=600, n_ctx=512)
    emb = TorchTokenEmbedder(shape, device="cuda", seed=9, n_batch=512)
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    gi = None
    all_chunks = []
    for doc in range(3):
        sents = make_sentences(500 + doc, 40)
        X = raglite_amd.embed_strings(sents, config=cfg, embedder=emb)  # (40, 128) fp16, unit rows
        chunks, chunk_embs = raglite_amd.split_chunks(sents, X, max_size=600)
        assert "".join(chunks) == "".join(sents) and sum(len(m) for m in chunk_embs) == 40
        assert all(len(c) <= 600 for c in chunks)
        ids = [f"doc{doc}-chunk{i}" for i in range(len(chunks))]
        if gi is None:
            gi = raglite_amd.GpuIndex(ids, chunk_embs, docs=chunks, metadata=[{"doc": doc}] * len(chunks), storage="f16")
        else:
            gi.insert_chunks(ids, chunk_embs, docs=chunks, metadata=[{"doc": doc}] * len(chunks))
        all_chunks += list(zip(ids, chunk_embs))
    # every chunk is found by one of its own chunklet embeddings, with similarity ~1
    for cid, m in all_chunks[::5]:
        got, sc = raglite_amd.vector_search(np.asarray(m[0]), num_results=1, config=cfg, index=gi)
        assert got == [cid] and abs(sc[0] - 1.0) < 2e-3
    only1, _ = raglite_amd.vector_search(np.asarray(all_chunks[0][1][0]), num_results=50, metadata_filter={"doc": 1},
                                         config=cfg, index=gi)
    assert only1 and all(c.startswith("doc1-") for c in only1)
    assert gi.delete_chunks([all_chunks[0][0]]) == 1
    got, _ = raglite_amd.vector_search(np.asarray(all_chunks[0][1][0]), num_results=3, config=cfg, index=gi)
    assert all_chunks[0][0] not in got
    # late-interaction rerank with the same encoder on the query side (`RAGLiteConfig.reranker`, `_search.py:364-397`):
    # the scores are the oracle's MaxSim of the query's token vectors against each candidate's chunklet vectors
    ranker = raglite_amd.MaxSimRanker.from_embedder(gi, emb)
    cfg_r = raglite_amd.HotPathConfig(vector_search_query_adapter=False, reranker=ranker)
    class _Chunk:  # the reference passes Chunk objects whose str() is the chunk text (`_search.py:395`)
        def __init__(self, text): self.text = text
        def __str__(self): return self.text

    cand = [_Chunk(gi.docs[i]) for i in (3, 7, 1, 9)]
    out = raglite_amd.rerank_chunks("Some question about the third document?", cand, config=cfg_r)
    assert sorted(c.text for c in out) == sorted(c.text for c in cand)
    qv = ranker.query_encoder("Some question about the third document?")
    E_all = np.vstack([np.asarray(m, dtype=np.float32) for _, m in all_chunks])
    off_all = np.concatenate(([0], np.cumsum([len(m) for _, m in all_chunks])))
    ref = oracle.maxsim_candidates(E_all, off_all, qv, [3, 7, 1, 9])
    np.testing.assert_allclose(ranker.score("Some question about the third document?", [3, 7, 1, 9]), ref, atol=1e-3)
    assert [cand.index(c) for c in out] == np.argsort(-ref, kind="stable").tolist()
    gi.close()


# ---------------------------------------------------------------------------------------------------
# hypothesis-generated ragged shapes (KAT class iv)
# ---------------------------------------------------------------------------------------------------
def test_fuzz_ragged_shapes_integer_exact():
    """Random CSR layouts (empty chunks, single-row chunks, long chunks), row counts around the 16-row tile and the
    workgroup-range boundaries, every dim class (stream fast path and generic fallback), 1..40 query vectors, both
    storages: MaxSim scores, exact top-k and the two-stage search are bit-identical to the oracle on integer data."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    import os

    n_examples = int(os.environ.get("RAGLITE_FUZZ_EXAMPLES", "150"))  # crank up for a soak run

    @settings(max_examples=n_examples, deadline=None, derandomize=n_examples <= 150, suppress_health_check=list(HealthCheck))
    @given(
        sizes=st.lists(st.one_of(st.just(0), st.integers(1, 3), st.integers(1, 40)), min_size=1, max_size=400),
        dim=st.sampled_from([128, 256, 384, 512, 768, 1024, 64, 100]),
        nq=st.integers(1, 40),
        storage=st.sampled_from(["f32", "f16"]),
        seed=st.integers(0, 10_000),
    )
    def run(sizes, dim, nq, storage, seed):
        if storage == "f16" and dim not in (128, 256, 384, 512, 768, 1024):
            storage = "f32"
        off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        n = int(off[-1])
        if n == 0:
            return
        E = oracle.synth_matrix(seed, n, dim, "small_int")
        Q = oracle.synth_matrix(seed + 1, nq, dim, "small_int")
        idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
        try:
            ref = oracle.maxsim_scores(E, off, Q).astype(np.float32)
            assert np.array_equal(idx.maxsim_scores(Q), ref)
            k = min(17, len(sizes))
            s, c = idx.maxsim_topk(Q, k)
            es, ec = oracle.topk_desc(ref, k)
            es, ec = np.where(np.isneginf(es), es, es), ec
            assert np.array_equal(s, es)
            assert np.array_equal(c[np.isfinite(s)], ec[np.isfinite(es)])
            r2c = np.repeat(np.arange(len(sizes)), sizes)
            gs, gc, cnt = idx.search_chunks(Q[0], 20, 5)
            ws, wc = oracle.search_chunks(E, r2c, Q[0], 20, 5, "dot", np.float32)
            assert int(cnt) == len(wc) and np.array_equal(gc[: len(wc)], wc) and np.array_equal(gs[: len(wc)], ws)
        finally:
            idx.close()

    run()


def test_adapter_apply_against_reference_lines():
    """a5 pinned: `(Q @ q).astype(q.dtype)` exec'd from vector_search's body (tests/golden/query_adapter.npz).  The
    device multiplies in fp32 (the reference in fp64): after the cast back to the query's fp16 the results agree to
    within one fp16 ulp, and exactly in the overwhelming majority of elements."""
    from pathlib import Path

    g = np.load(Path(__file__).parent / "golden" / "query_adapter.npz")
    for i in range(int(g["n_a5_cases"])):
        A, q, want = g[f"a5_{i}_A"], g[f"a5_{i}_q"], g[f"a5_{i}_out"]
        if want.dtype == np.float16:
            got = raglite_amd.adapter_apply(A.astype(np.float32), q.astype(np.float32), want_f16=True)
            ulps = np.abs(got.view(np.int16).astype(np.int32) - want.view(np.int16).astype(np.int32))
            assert ulps.max() <= 1 and (ulps == 0).mean() > 0.97
        else:
            got = raglite_amd.adapter_apply(A.astype(np.float32), q.astype(np.float32))
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("B", [7, 40, 130])
def test_l2_batched_near_duplicates_are_rescored(storage, B):
    """The batched paths rank l2 by |e|^2 + |q|^2 - 2 e.q, which cancels for near-duplicates of large norm; the k hits
    are re-scored with the exact sum (e - q)^2, so even a query that IS a stored row (plus fp16 rounding noise) reports
    its true similarity within 1e-4 -- for every batch class, also under a filter."""
    n, dim = 3000, 1024
    E = (3.0 * oracle.synth_matrix(121, n, dim)).astype(np.float16).astype(np.float32)  # |e|^2 ~ 3000: worst case
    rng = np.random.default_rng(16)
    picks = rng.choice(n, B, replace=False)
    Q = E[picks] + (1e-3 * oracle.synth_matrix(122, B, dim)).astype(np.float32)  # near-duplicates, true distance ~0.02
    idx = raglite_amd.DeviceIndex(E, metric="l2", storage=storage)
    S, R = idx.search_rows(Q, 20)
    for b in (0, B // 2, B - 1):
        ref = oracle.similarity(E, Q[b], "l2")
        assert R[b, 0] == picks[b]
        assert_topk_close(S[b], R[b], ref, 20, TOL)
    flt = np.ones(n, bool)
    flt[picks[0]] = False  # one row per chunk: the best hit of query 0 is filtered away
    S2, R2 = idx.search_rows(Q, 20, chunk_filter=flt)
    assert picks[0] not in R2[0]
    assert_topk_close(S2[0], R2[0], np.where(flt, oracle.similarity(E, Q[0], "l2"), -np.inf), 20, TOL)
    idx.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_lifecycle_random_operation_sequences(storage):
    """Random interleavings of append / delete / filtered and unfiltered searches against a NumPy mirror of the table
    (integer data: bit-exact).  Exercises capacity growth (several reallocations), tombstones that are extended by
    appends, deletes of already deleted chunks, filters over deleted chunks and k larger than what is left."""
    rng = np.random.default_rng(20 if storage == "f32" else 21)
    dim = 256
    E = oracle.synth_matrix(200, 40, dim, "small_int")
    sizes = [int(x) for x in rng.integers(1, 6, size=12)]
    sizes[-1] += 40 - sum(sizes) if sum(sizes) < 40 else 0
    while sum(sizes) > 40:
        sizes.pop()
    sizes.append(40 - sum(sizes)) if sum(sizes) < 40 else None
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
    alive = np.ones(len(off) - 1, bool)
    seed = 300
    for step in range(40):
        op = rng.choice(["append", "delete", "search", "filter", "maxsim"], p=[0.25, 0.2, 0.2, 0.2, 0.15])
        n_chunks = len(off) - 1
        r2c = np.repeat(np.arange(n_chunks), np.diff(off))
        if op == "append":
            n_new_chunks = int(rng.integers(1, 8))
            new_sizes = rng.integers(0 if step % 7 == 0 else 1, 9, size=n_new_chunks)
            rows = oracle.synth_matrix(seed, int(new_sizes.sum()), dim, "small_int")
            seed += 1
            idx.append(rows, new_sizes)
            E = np.concatenate([E, rows])
            off = np.concatenate([off, off[-1] + np.cumsum(new_sizes)]).astype(np.int64)
            alive = np.concatenate([alive, np.ones(n_new_chunks, bool)])
            assert idx.n_rows == len(E) and idx.n_chunks == len(off) - 1
        elif op == "delete":
            dead = rng.choice(n_chunks, size=int(rng.integers(1, max(2, n_chunks // 6))), replace=False)
            idx.delete_chunks(dead)
            alive[dead] = False
            assert idx.live() == (int(alive[r2c].sum()), int(alive.sum()))
        else:
            q = oracle.synth_matrix(seed, 3, dim, "small_int")
            seed += 1
            flt = (rng.random(n_chunks) < 0.6) if op == "filter" else None
            ok = alive if flt is None else alive & flt
            if op == "maxsim":
                ws, wc = oracle.maxsim_topk_filtered(E, off, q, 12, ok)
                gs, gc = idx.maxsim_topk(q, 12, chunk_filter=flt)
                assert np.array_equal(gc, wc) and np.array_equal(gs, ws.astype(np.float32))
            else:
                for b in range(3):
                    ws, wr = oracle.search_rows_filtered(E, r2c, q[b], 25, ok, "dot", np.float64)
                    gs, gr = idx.search_rows(q[b], 25, chunk_filter=flt)
                    assert np.array_equal(gr, wr), (step, op)
                    assert np.array_equal(gs, np.where(wr >= 0, ws, -np.inf).astype(np.float32))
    idx.close()


def test_pure_c_consumer(tmp_path):
    """The drop-in boundary is a C ABI: tests/c/abi_smoke.c includes only include/raglite_hip.h, is compiled with gcc
    and linked against libraglite_hip.so -- no Python, torch or HIP headers -- and checks known answers."""
    import shutil
    import subprocess
    from pathlib import Path

    from raglite_amd import _build

    root = Path(__file__).resolve().parent.parent
    lib = _build.LIB_PATH
    assert lib.exists(), "build the library first (python -m raglite_amd._build)"
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = tmp_path / "abi_smoke"
    subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", f"-I{root / 'include'}", str(root / "tests" / "c" / "abi_smoke.c"),
                    "-o", str(exe), f"-L{lib.parent}", "-lraglite_hip", "-lm", f"-Wl,-rpath,{lib.parent}"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "abi_smoke OK" in res.stdout


def test_encoders_match_huggingface_on_device(torch_cuda):
    """8f-2 / 8f-4 on the GPU: the token encoder and the cross-encoder, in fp32 on cuda:0, against Hugging Face's CPU
    fp32 `XLMRobertaModel` / `BertForSequenceClassification` with the same weights (tolerance 2e-4: hipBLASLt and the
    SDPA kernel accumulate in another order); bf16, the serving dtype, must reproduce the fp32 ranking up to pairs
    whose scores differ by less than bf16 noise."""
    import torch
    from transformers import BertConfig, BertForSequenceClassification, XLMRobertaConfig, XLMRobertaModel

    from raglite_amd._cross_encoder import CrossEncoderShape, TorchCrossEncoderRanker
    from raglite_amd._torch_embedder import EncoderShape, TorchTokenEmbedder

    torch.manual_seed(21)
    xcfg = XLMRobertaConfig(vocab_size=2000, hidden_size=128, num_hidden_layers=3, num_attention_heads=8, intermediate_size=256,
                            max_position_embeddings=260, type_vocab_size=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.0,
                            attention_probs_dropout_prob=0.0, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    hf = XLMRobertaModel(xcfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    shape = EncoderShape(vocab_size=2000, hidden=128, layers=3, heads=8, ffn=256, max_positions=260, n_ctx=256)
    emb = TorchTokenEmbedder(shape, device="cuda", dtype=torch.float32)
    emb.load_hf_state_dict(hf.state_dict())
    g = torch.Generator().manual_seed(6)
    rows = [[0, *torch.randint(3, 2000, (n,), generator=g).tolist(), 2] for n in (200, 31, 254, 1, 77)]
    T = max(len(r) for r in rows)  # noqa: N806
    ids = torch.full((len(rows), T), 1, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = torch.tensor(r)
    lengths = torch.tensor([len(r) for r in rows])
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=(torch.arange(T)[None, :] < lengths[:, None]).long()).last_hidden_state
        got = emb.encoder(ids.cuda(), lengths.cuda()).cpu()
    for i, r in enumerate(rows):
        torch.testing.assert_close(got[i, : len(r)], want[i, : len(r)], atol=2e-4, rtol=1e-4)

    bcfg = BertConfig(vocab_size=3000, hidden_size=128, num_hidden_layers=3, num_attention_heads=8, intermediate_size=256,
                      max_position_embeddings=128, type_vocab_size=2, num_labels=1, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12)
    hb = BertForSequenceClassification(bcfg).eval()
    with torch.no_grad():
        for p in hb.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cshape = CrossEncoderShape(vocab_size=3000, hidden=128, layers=3, heads=8, ffn=256, max_positions=128, n_ctx=128)
    rk = TorchCrossEncoderRanker(cshape, device="cuda", dtype=torch.float32, pairs_per_batch=16)
    rk.load_hf_state_dict(hb.state_dict())
    query = "how does the reranker order passages"
    docs = make_sentences(91, 50)
    docs[7] = ""
    docs[9] = docs[9] * 30  # truncated
    pairs = rk.encode_pairs(query, docs)
    T = max(len(p[0]) for p in pairs)  # noqa: N806
    ids = torch.zeros((len(pairs), T), dtype=torch.long)
    for i, p in enumerate(pairs):
        ids[i, : len(p[0])] = torch.tensor(p[0])
    lengths = torch.tensor([len(p[0]) for p in pairs])
    first = torch.tensor([p[1] for p in pairs])
    ar = torch.arange(T)
    with torch.no_grad():
        want = hb(input_ids=ids, attention_mask=(ar[None, :] < lengths[:, None]).long(),
                  token_type_ids=((ar[None, :] >= first[:, None]) & (ar[None, :] < lengths[:, None])).long()).logits[:, 0]
    got = rk.logits(query, docs)
    assert got.is_cuda
    torch.testing.assert_close(got.cpu(), want, atol=2e-4, rtol=1e-4)
    res = rk.rank(query=query, docs=docs)
    sig = torch.sigmoid(want).numpy()
    got_order = [r.doc_id for r in res.results]
    for a, b in zip(got_order, got_order[1:]):  # best first w.r.t. the HF scores, up to fp32 noise
        assert sig[a] >= sig[b] - 1e-5
    # bf16 serving dtype: same weights, ranking agrees wherever the fp32 scores are separated by more than bf16 noise
    rk16 = TorchCrossEncoderRanker(cshape, device="cuda", dtype=torch.bfloat16)
    rk16.load_hf_state_dict(hb.state_dict())
    z16 = rk16.logits(query, docs).cpu()
    assert torch.isfinite(z16).all()
    scale = float(want.abs().max()) + 1.0
    assert float((z16 - want).abs().max()) < 0.08 * scale


# ---------------------------------------------------------------------------------------------------
# Arithmetic of the MFMA streaming kernel: fp16 (hi, lo) split of fp32 operands vs the exact fp32 chain
# ---------------------------------------------------------------------------------------------------
def _maxsim_truth(E32, Q32, off):
    S = E32.astype(np.float64) @ Q32.astype(np.float64).T  # noqa: N806
    out = np.full(len(off) - 1, -np.inf)
    for c in range(len(off) - 1):
        if off[c + 1] > off[c]:
            out[c] = S[off[c] : off[c + 1]].max(axis=0).sum()
    return out, (np.abs(E32.astype(np.float64)) @ np.abs(Q32.astype(np.float64)).T).sum(axis=1).max()


@pytest.mark.parametrize("kind", ["normalized", "uniform", "scaled_down", "scaled_up", "sparse_tiny"])
def test_split_arithmetic_is_as_accurate_as_fp32(kind):
    """The default arithmetic of the streaming kernel writes every fp32 operand as an fp16 (hi, lo) pair (22 bits) and
    multiplies on the fp16 matrix pipe with fp32 accumulation (include/raglite_hip.h, rl_index_set_arithmetic).  Against
    float64 truth its error must stay within 2x that of the exact fp32 MFMA chain (+ one fp32 ulp of the largest
    |e||q| sum), for corpora and queries at any power-of-two-ish scale -- both are rescaled inside the kernel."""
    rng = np.random.default_rng(31)
    n, dim, nq = 6000, 1024, 32
    E = rng.standard_normal((n, dim))
    Q = rng.standard_normal((nq, dim))
    if kind == "normalized":
        E /= np.linalg.norm(E, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    elif kind == "uniform":
        E, Q = rng.uniform(-1, 1, (n, dim)), rng.uniform(-1, 1, (nq, dim))
    elif kind == "scaled_down":
        E *= 3e-21
        Q *= 7e-12
    elif kind == "scaled_up":
        E *= 5e17
        Q *= 9e9
    elif kind == "sparse_tiny":  # elements spanning many orders of magnitude inside every row
        E *= 10.0 ** rng.uniform(-8, 0, (n, dim))
        Q *= 10.0 ** rng.uniform(-5, 0, (nq, dim))
    E32, Q32 = E.astype(np.float32), Q.astype(np.float32)
    off = ragged_offsets(rng, n, 1, 9)
    truth, unit = _maxsim_truth(E32, Q32, off)
    idx = raglite_amd.DeviceIndex(E32, off, metric="dot")
    assert idx.arithmetic == "f16_split"
    split = np.asarray(idx.maxsim_scores(Q32), dtype=np.float64)
    rows_split = np.asarray(idx.search_rows(Q32[:7], 5)[0], dtype=np.float64)  # mode 1 (row scores) of the same kernel
    idx.set_exact_fp32()
    assert idx.arithmetic == "fp32_exact"
    exact = np.asarray(idx.maxsim_scores(Q32), dtype=np.float64)
    rows_exact = np.asarray(idx.search_rows(Q32[:7], 5)[0], dtype=np.float64)
    idx.set_exact_fp32(False)
    assert idx.arithmetic == "f16_split"
    live = np.isfinite(truth)
    err_split, err_exact = np.abs(split - truth)[live].max(), np.abs(exact - truth)[live].max()
    assert err_split <= 2.0 * err_exact + 32 * 2.0 ** -24 * unit, (err_split, err_exact, unit)
    assert err_exact <= 32 * 1e-6 * unit
    np.testing.assert_allclose(rows_split, rows_exact, rtol=2e-5, atol=1e-6 * unit / 32)
    idx.close()


def test_split_arithmetic_eligibility(torch_cuda):
    """AUTO picks the split only where it cannot lose precision: all rows finite and all non-zero row norms within 2^10
    of each other; an append that breaks this switches the index to the exact chain; integer data stays bit-exact."""
    rng = np.random.default_rng(32)
    n, dim = 2048, 256
    E = rng.standard_normal((n, dim)).astype(np.float32)
    off = np.arange(0, n + 1, 4, dtype=np.int64)
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    assert idx.arithmetic == "f16_split"
    idx.append(np.zeros((4, dim), np.float32), [4])  # all-zero rows do not count
    assert idx.arithmetic == "f16_split"
    idx.append((E[:4] * 3000.0).astype(np.float32), [4])  # norm ratio 3000 > 2^10
    assert idx.arithmetic == "fp32_exact"
    idx.close()
    wide = E.copy()
    wide[5] *= 1e-5
    idx = raglite_amd.DeviceIndex(wide, off, metric="cosine")
    assert idx.arithmetic == "fp32_exact"
    idx.close()
    bad = E.copy()
    bad[9, 3] = np.inf
    idx = raglite_amd.DeviceIndex(bad, off, metric="dot")
    assert idx.arithmetic == "fp32_exact"
    idx.close()
    idx = raglite_amd.DeviceIndex(E.astype(np.float16), off, metric="dot", storage="f16")
    assert idx.arithmetic == "f16_stored"
    idx.close()
    # integers up to 2^21 are exact in a (hi, lo) pair: scores are bit-identical to integer arithmetic
    Ei = rng.integers(-1500, 1500, size=(n, dim)).astype(np.float32)
    Qi = rng.integers(-3, 4, size=(9, dim)).astype(np.float32)
    idx = raglite_amd.DeviceIndex(torch_cuda.from_numpy(Ei).cuda(), off, metric="dot")
    assert idx.arithmetic == "f16_split"
    got = idx.maxsim_scores(torch_cuda.from_numpy(Qi).cuda()).cpu().numpy()
    S = Ei.astype(np.int64) @ Qi.astype(np.int64).T  # noqa: N806
    want = S.reshape(n // 4, 4, 9).max(axis=1).sum(axis=1)
    assert np.array_equal(got.astype(np.int64), want) and np.array_equal(got, want.astype(np.float32))
    idx.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("dim", [128, 256, 384, 512, 768, 1024])
@pytest.mark.parametrize("n_rows,nq", [(20, 32), (700, 17), (9000, 32), (40_000, 25)])
def test_maxsim_batch_two_queries_per_pass(dim, n_rows, nq, storage):
    """`rl_maxsim_topk_batch` scores two queries per corpus pass (maxsim_stream2_kernel: eight symmetric waves, query
    group x K-quarter) wherever the fp16-split arithmetic is in effect and a query has 17..32 vectors.  Integer data:
    scores are exact, so batch == one-query-at-a-time == oracle, bit for bit, incl. the odd query left over, empty
    chunks, tombstones and workgroups with 1, 2, 3 or many tiles."""
    rng = np.random.default_rng(dim + n_rows)
    off = ragged_offsets(rng, n_rows, 1, 15, empty_every=23)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(900 + dim, n_rows, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(950 + i, nq, dim, "small_int") for i in range(5)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)  # small integers are exact in fp16 storage too
    assert idx.arithmetic == ("f16_split" if storage == "f32" else "f16_stored")
    k = min(50, n_chunks)
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in range(5):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        ss, sc = idx.maxsim_topk(Qb[i], k)
        assert np.array_equal(bc[i][: len(wc)], wc) and np.array_equal(bs[i][: len(wc)], ws)
        assert np.array_equal(bc[i], sc) and np.array_equal(bs[i], ss)
    if n_chunks > 8:
        dead = rng.choice(n_chunks, size=n_chunks // 4, replace=False)
        idx.delete_chunks(dead)
        bs, bc = idx.maxsim_topk_batch(Qb[:4], k)
        for i in range(4):
            ss, sc = idx.maxsim_topk(Qb[i], k)
            assert np.array_equal(bc[i], sc) and np.array_equal(bs[i], ss)
            assert not np.isin(bc[i][bc[i] >= 0], dead).any()
    idx.close()


@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_maxsim_batch_pairs_float_data_device_pointers(torch_cuda, storage):
    """Float data, CUDA tensors: the pair kernel's scores equal the single-query kernel's bit for bit (same arithmetic,
    same K order), and the oracle's within tolerance; fp32- and fp16-stored corpus."""
    torch = torch_cuda
    n, dim, nq = 30_000, 1024, 32
    rng = np.random.default_rng(77)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=31)
    Qb = torch.empty((6, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Qb, seed=32)
    if storage == "f16":
        E = E.half()
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
    Eh = E.float().cpu().numpy()
    # batches of three or more take the eight-query kernel over the corpus image (tests/test_gpu_gemm_pass.py); pairs here
    parts = [idx.maxsim_topk_batch(Qb[i : i + 2], 100) for i in (0, 2, 4)]
    bs, bc = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    for i in range(6):
        ss, sc = idx.maxsim_topk(Qb[i], 100)
        assert torch.equal(bc[i], sc) and torch.equal(bs[i], ss)
    ws, wc = oracle.maxsim_topk(Eh, off, Qb[0].cpu().numpy(), 100, np.float64)
    assert set(wc.tolist()) == set(bc[0].cpu().numpy().tolist())
    np.testing.assert_allclose(np.sort(bs[0].cpu().numpy())[::-1], np.sort(ws)[::-1], rtol=0, atol=2e-4)
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_split_arithmetic_gemm_and_rerank_query_scales(metric):
    """Every query gets a power-of-two scale of its own in the fp16-split GEMM (query_presplit_kernel) and in the rerank
    kernel: a batch whose queries span 20 orders of magnitude must score each query as accurately as it would alone --
    cosine is scale-free (compare with float64 directly), dot scores are compared relative to |q|."""
    rng = np.random.default_rng(41)
    n, dim, B = 3000, 128, 130  # dim 128: the rerank MFMA kernel applies; B >= 96: the GEMM path
    E = rng.standard_normal((n, dim)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    mag = 10.0 ** rng.uniform(-10, 10, size=B)
    Q = (rng.standard_normal((B, dim)) * mag[:, None]).astype(np.float32)
    off = np.arange(0, n + 1, 4, dtype=np.int64)
    idx = raglite_amd.DeviceIndex(E, off, metric=metric)
    assert idx.arithmetic == "f16_split"
    s, r = idx.search_rows(Q, 10)
    E64, Q64 = E.astype(np.float64), Q.astype(np.float64)
    dots = Q64 @ E64.T
    ref = dots / (np.linalg.norm(Q64, axis=1)[:, None] * np.linalg.norm(E64, axis=1)[None, :]) if metric == "cosine" else 1.0 + dots
    unit = np.ones(B) if metric == "cosine" else np.maximum(np.linalg.norm(Q64, axis=1), 1.0)
    for b in range(B):
        want = np.sort(ref[b])[::-1][:10]
        np.testing.assert_allclose(np.asarray(s[b], dtype=np.float64) / unit[b], want / unit[b], rtol=0, atol=2e-5)
    # rerank kernel: 32 query vectors per query, each query at its own magnitude
    nq = 32
    Qv = (rng.standard_normal((6, nq, dim)) * (10.0 ** rng.uniform(-8, 8, size=6))[:, None, None]).astype(np.float32)
    cand = rng.integers(0, n // 4, size=(6, 40)).astype(np.int32)
    got = np.asarray(idx.maxsim_rerank(Qv, cand), dtype=np.float64)
    for qi in range(6):
        S = Qv[qi].astype(np.float64) @ E64.T  # noqa: N806
        want = np.array([S[:, 4 * c : 4 * c + 4].max(axis=1).sum() for c in cand[qi]])
        scale = np.abs(Qv[qi].astype(np.float64)).max() * nq
        np.testing.assert_allclose(got[qi] / scale, want / scale, rtol=0, atol=2e-6)
    idx.close()


@pytest.mark.parametrize("dim", [128, 96])  # 128: the MFMA candidate kernel; 96: the generic one-wave-per-item kernel
def test_maxsim_rerank_unscorable_candidates_are_minus_inf(torch_cuda, dim):
    """`search_chunks` pads its chunk ordinals with -1 and device callers are not validated on the host: a candidate of
    -1, an ordinal outside [0, n_chunks) and a tombstoned chunk all score -inf (no out-of-bounds read), everything else
    is unchanged -- host and device arguments alike."""
    torch = torch_cuda
    rng = np.random.default_rng(dim)
    n, nq = 900, 20
    off = ragged_offsets(rng, n, 1, 9)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(70, n, dim, "small_int")
    Q = np.stack([oracle.synth_matrix(71 + i, nq, dim, "small_int") for i in range(3)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    cand = rng.integers(0, n_chunks, size=(3, 24)).astype(np.int32)
    want = np.asarray(idx.maxsim_rerank(Q, cand)).copy()
    dead = np.unique(cand[:, 5])
    idx.delete_chunks(dead)
    bad = cand.copy()
    bad[:, 0] = -1
    got_host = np.asarray(idx.maxsim_rerank(Q, bad))
    bad_dev = bad.copy()
    bad_dev[:, 1] = n_chunks + 12345  # only a device caller can hand this over
    bad_dev[:, 2] = -7
    got_dev = idx.maxsim_rerank(torch.as_tensor(Q, device="cuda"), torch.as_tensor(bad_dev, device="cuda")).cpu().numpy()
    gone = np.isin(cand, dead)
    for got, mask in ((got_host, gone | (np.arange(24) == 0)[None, :]), (got_dev, gone | (np.arange(24) < 3)[None, :])):
        assert np.all(np.isneginf(got[mask]))
        assert np.array_equal(got[~mask], want[~mask])
    with pytest.raises(ValueError):
        idx.maxsim_rerank(Q, np.full((3, 4), n_chunks, dtype=np.int32))  # host callers are still validated
    idx.close()


def test_one_handle_on_two_streams_is_serialised(torch_cuda):
    """The index' score / selection scratch is shared by all calls on a handle: calls arriving on different streams
    must not overlap on the device (the second one waits for the first stream).  Alternating two streams with
    different queries has to give each query its own result."""
    torch = torch_cuda
    n, dim = 200_000, 256
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=91, kind="small_int")
    idx = raglite_amd.DeviceIndex(E, None, metric="dot")
    Q = torch.empty((8, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=92, kind="small_int")
    torch.cuda.synchronize()
    want = [tuple(t.clone() for t in idx.search_rows(Q[i], 50)) for i in range(8)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for i in range(8):
        with torch.cuda.stream(streams[i % 2]):
            got.append(idx.search_rows(Q[i], 50))
    torch.cuda.synchronize()
    for i in range(8):
        assert torch.equal(got[i][1], want[i][1]) and torch.equal(got[i][0], want[i][0])
    idx.close()


def test_vector_search_raises_beyond_the_exact_topk_limit():
    """`num_hits` / `num_results` beyond the 2048 rows the exact selection ranks raise instead of being clamped."""
    rng = np.random.default_rng(3)
    mats = [rng.standard_normal((2, 32)).astype(np.float16) for _ in range(50)]
    gi = raglite_amd.GpuIndex([f"c{i}" for i in range(50)], mats, metric="cosine")
    q = rng.standard_normal(32).astype(np.float16)
    ids, scores = raglite_amd.vector_search(q, num_results=512, index=gi)  # 4 * 512 = 2048 rows: the limit itself
    assert len(ids) == 50
    with pytest.raises(ValueError, match="exact top-k"):
        raglite_amd.vector_search(q, num_results=513, index=gi)
    gi.close()
