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
    with idx.options(hi_maxsim=0):
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
    # lazy images: the first batch has rows + HI image, its fallback runs the streaming kernels; it leaves word that it fell back, and the
    # second batch asks for the pre-split image: from then on the fallback is the eight-query full-precision pass
    s1, c1 = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"] and idx.memory()["presplit_image"] == 0
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"] and idx.memory()["presplit_image"] > 0
    with idx.options(hi_maxsim=0):
        fs, fc = idx.maxsim_topk_batch(Qb, 100)
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    assert np.isin(bc, hot).all()
    # (the streaming kernels sum in another order, and 4 000 chunks sit within 1e-4 of each other: which of them make the top-100 is decided
    # in the last bits -- hold the first batch to float64 with the tie-aware check, like every float-data result)
    for i in range(len(Qb)):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        assert_topk_close(s1[i], c1[i], ref, 100, 2e-6 * float(np.abs(ref).max()))
        assert np.isin(c1[i], hot).all()
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
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    with idx.options(hi_products=1 if one == "1" else 2, pp_pass=0 if nopp == "1" else 1):
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
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    with idx.options(hi_products=1 if one == "1" else 2, pp_pass=0 if nopp == "1" else 1):
        bs, bc = idx.maxsim_topk_batch(Qb, k)
        s1, r1 = idx.search_rows(Qb[0, 0], 50)  # the single-query half-bytes search reads the same halves and norms
    with idx.options(hi_maxsim=0, hi_search=0):
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
    idx.set_option("lazy_images", 0)  # (with the pre-split image from the start the guarded fallback IS the full-precision pass below)
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"]
    with idx.options(hi_maxsim=0):
        fs, fc = idx.maxsim_topk_batch(Qb, 100)
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    idx.close()


def _torch():
    import torch

    raglite_amd.set_device(0)
    return torch


def _corpus(torch, n, dim, seed, kind="uniform"):
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=seed, kind=kind)
    return E


def _queries(torch, n_queries, nq, dim, seed, kind="uniform"):
    Q = torch.empty((n_queries, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=seed, kind=kind)
    return Q


@pytest.mark.parametrize("kind,n_queries", [("small_int", 16), ("uniform", 21), ("uniform", 8)])
def test_sharded_batch_with_one_global_threshold_equals_single_index(kind, n_queries):
    """`rl_maxsim_batch_begin` / `_finish` (driven by `ShardedIndex._local_maxsim_batch`): three shards of one corpus exchange their k best
    approximate scores and bounds -- stacked in plain Python here instead of an all-gather -- and re-score only what could be in the GLOBAL
    top-k.  The merge of their lists is what ONE index over the whole corpus returns (`/root/reference/src/raglite/_search.py:143-149`
    semantics, MaxSim generalisation behind the reranker plugin call `:394-396`): bit for bit on integer data, the same chunks with
    scores to the last bits of the fp32 sums on float data (both come from the same exact re-scoring kernel).  And the shards together
    re-score far fewer candidates than each would alone."""
    torch = _torch()
    n, dim, nq, k = 210_000, 1024, 32, 50
    rng = np.random.default_rng(23)
    off = ragged_offsets(rng, n, 1, 15)
    E = _corpus(torch, n, dim, seed=77, kind=kind)
    Q = _queries(torch, n_queries, nq, dim, seed=78, kind=kind)
    whole = raglite_amd.DeviceIndex(E, off, metric="dot")
    ws, wc = whole.maxsim_topk_batch(Q, k)
    assert whole.filter_stats()["kind"] == "maxsim_batch_hi" and not whole.filter_stats()["fallback"]
    n_chunks = len(off) - 1
    cuts = [0, n_chunks // 3, 2 * n_chunks // 3 + 5, n_chunks]
    shards = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(off[lo]), int(off[hi])
        shards.append((raglite_amd.DeviceIndex(E[r0:r1].clone(), off[lo : hi + 1] - off[lo], metric="dot"), lo))
    approx = [sh.maxsim_batch_begin(Q, k) for sh, _ in shards]
    for a in approx:  # k best approximate scores, descending, then a positive bound
        assert tuple(a.shape) == (n_queries, k + 1) and bool((a[:, : k - 1] >= a[:, 1:k]).all()) and bool((a[:, k] > 0).all())
    allg = torch.stack(approx)
    lists_s, lists_c, staged_candidates = [], [], 0.0
    for r, (sh, lo) in enumerate(shards):
        s, c = sh.maxsim_batch_finish(Q, allg, r, k)
        st = sh.filter_stats()
        assert st["kind"] == "maxsim_batch_hi" and not st["fallback"]
        staged_candidates += st["candidates_per_query_mean"]
        lists_s.append(s)
        lists_c.append(torch.where(c >= 0, c + lo, torch.full_like(c, -1)))
    ms, mc = raglite_amd.merge_topk(torch.stack(lists_s), torch.stack(lists_c).to(torch.int32), k)
    assert torch.equal(mc.to(torch.int64), wc.to(torch.int64))
    if kind == "small_int":
        assert torch.equal(ms, ws)
    else:
        assert float((ms - ws).abs().max()) <= 2e-6 * float(ws.abs().max())
    alone = 0.0
    for sh, _ in shards:  # what the shards re-score without the exchange
        sh.maxsim_topk_batch(Q, k)
        alone += sh.filter_stats()["candidates_per_query_mean"]
    # like for like: the shards share ONE threshold derived from approximate scores, so the yardstick is the single index with its
    # approximate threshold (its default since round 4, the second threshold from the exact scores of its approximate top-k, re-scores
    # fewer still: that one needs no exchange on a single index and would need a third one across shards)
    with whole.options(exact_kth_threshold=0):
        whole.maxsim_topk_batch(Q, k)
        single = whole.filter_stats()["candidates_per_query_mean"]
    assert staged_candidates <= 1.3 * single + 3 and staged_candidates < 0.75 * alone, (staged_candidates, single, alone)
    for i in [whole, *[sh for sh, _ in shards]]:
        i.close()


def test_batch_begin_refuses_what_the_pipeline_does_not_cover():
    torch = _torch()
    from raglite_amd._abi import UnsupportedError

    small = raglite_amd.DeviceIndex(_corpus(torch, 5_000, 256, seed=5), None, metric="dot")  # no image of the hi halves
    with pytest.raises(UnsupportedError):
        small.maxsim_batch_begin(_queries(torch, 8, 4, 256, seed=6), 10)
    big = raglite_amd.DeviceIndex(_corpus(torch, 70_000, 1024, seed=7), None, metric="dot")
    with pytest.raises(UnsupportedError):
        big.maxsim_batch_begin(_queries(torch, 2, 4, 1024, seed=8), 10)  # two queries: the pair kernel's business
    with pytest.raises(ValueError):
        big.maxsim_batch_finish(_queries(torch, 8, 4, 1024, seed=8), torch.zeros((2, 8, 11), device="cuda"), 0, 10)  # no begin in progress
    q8 = _queries(torch, 8, 4, 1024, seed=8)
    a = big.maxsim_batch_begin(q8, 10)
    big.search_rows(q8[:, 0, :].contiguous(), 5)  # another call on the index: the approximate scores of the batch are gone
    with pytest.raises(ValueError, match="another call"):
        big.maxsim_batch_finish(q8, torch.stack([a, a]), 0, 10)
    a = big.maxsim_batch_begin(q8, 10)
    s1, c1 = big.maxsim_batch_finish(q8, a[None], 0, 10)  # a world of one: the plain batch
    s0, c0 = big.maxsim_topk_batch(q8, 10)
    assert torch.equal(c0.to(torch.int64), c1.to(torch.int64)) and torch.equal(s0, s1)
    small.close()
    big.close()


# ---- fp16-STORED corpus (what RAGLite keeps: `/root/reference/src/raglite/_embed.py:140`, pgvector halfvec `_typing.py:211-232`): the same
# pipeline with the stored halves as the image of the approximate pass -- one fp16 MFMA product per multiply, q_hi . e; the bound has no
# e_lo term; the candidates are re-scored over the stored rows (fp16 -> fp32 on the way into v_mfma_f32_16x16x4_f32)
@pytest.mark.parametrize("n_queries,nq,k", [(16, 32, 100), (11, 17, 20)])
def test_f16_stored_batch_integer_bit_exact(n_queries, nq, k):
    rng = np.random.default_rng(n_queries)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(12_000 + nq, N, DIM, "small_int")  # exact in fp16
    Qb = np.stack([oracle.synth_matrix(12_100 + i, nq, DIM, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E.astype(np.float16), off, metric="dot", storage="f16")
    assert idx.arithmetic == "f16_stored" and idx.memory()["presplit_image"] == 0  # (lazy images: the first batch builds it)
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    assert idx.memory()["hi_image"] == 0 and idx.memory()["presplit_image"] > 0
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and st["queries"] == n_queries
    for i in sorted({0, n_queries // 2, n_queries - 1}):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i], ws)
    with idx.options(hi_maxsim=0):  # the eight-query two-product passes this replaces: the same bits on integer data
        s2, c2 = idx.maxsim_topk_batch(Qb, k)
    assert np.array_equal(bc, c2) and np.array_equal(bs, s2)
    idx.close()


def test_f16_stored_batch_float_data_bound_append_and_approx_scores():
    torch = _torch()
    n, nq, n_queries, k = 72_000, 32, 16, 100
    rng = np.random.default_rng(9)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.nn.functional.normalize(_corpus(torch, n, DIM, seed=91), dim=1).half()  # unit-norm rows rounded to fp16: what RAGLite stores
    Q = torch.nn.functional.normalize(_queries(torch, n_queries, nq, DIM, seed=92), dim=2)
    cut = len(off) // 2
    idx = raglite_amd.DeviceIndex(E[: int(off[cut])].clone(), off[: cut + 1], metric="dot", storage="f16")
    idx.append(E[int(off[cut]) :].float(), np.diff(off[cut:]))  # (the row-norm maximum follows the index)
    s, c = idx.maxsim_topk_batch(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and not st["fallback"] and k <= st["candidates_per_query_max"] < 1500
    # every score against float64 over the STORED values, the chunk sets tie-aware
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import bench_configs

    ref = bench_configs.maxsim_scores_f64(E.float(), off, Q)
    ref_top = ref.topk(k, dim=1)
    assert float((s.double() - ref_top.values).abs().max()) <= 2e-6 * float(ref_top.values.abs().max())
    kth = ref_top.values[:, -1:]
    got_ref = torch.gather(ref, 1, c.to(torch.int64))
    assert bool((got_ref >= kth - 1e-6).all())
    # the approximate scores stay inside the bound for EVERY chunk
    a, m = idx.maxsim_approx_scores(Q, kernel=0)
    err = (a.double() - ref).abs().max(dim=1).values
    assert bool((err <= m.double()).all()) and bool((m > 0).all())
    with idx.options(hi_maxsim=0):
        s2, c2 = idx.maxsim_topk_batch(Q, k)
    assert torch.equal(c.to(torch.int64), c2.to(torch.int64))
    assert float((s - s2).abs().max()) <= 2e-6 * float(s.abs().max())
    idx.close()



# ---- round 4: the second, tighter candidate threshold (option exact_kth_threshold) and the option plumbing ------------------------------------
@pytest.mark.parametrize("storage", ["f32", "f16"])
def test_exact_kth_threshold_same_bits_fewer_candidates(storage):
    """The approximate top-k is scored exactly first; the k-th best of THOSE scores, minus m, replaces (k-th approximate) - 2 m as the
    candidate threshold.  The result cannot move (both windows contain the exact top-k, the scores come from the same kernel); the
    number of chunks scored exactly goes down."""
    import torch

    raglite_amd.set_device(0)
    n, dim, nq, n_queries, k = 70_000, 1024, 32, 24, 100
    rng = np.random.default_rng(91)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=9100)
    Q = torch.empty((n_queries, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=9101)
    idx = raglite_amd.DeviceIndex(E.half() if storage == "f16" else E, off, metric="dot", storage=storage)
    assert idx.get_option("exact_kth_threshold") == 1
    s1, c1 = idx.maxsim_topk_batch(Q, k)
    st1 = idx.filter_stats()
    with idx.options(exact_kth_threshold=0):
        s0, c0 = idx.maxsim_topk_batch(Q, k)
        st0 = idx.filter_stats()
    assert st1["kind"] == st0["kind"] == "maxsim_batch_hi" and not st1["fallback"] and not st0["fallback"]
    assert torch.equal(c1, c0) and torch.equal(s1, s0)
    assert k <= st1["candidates_per_query_mean"] < st0["candidates_per_query_mean"], (st1, st0)
    print(f"[{storage}] candidates per query: {st0['candidates_per_query_mean']:.0f} -> {st1['candidates_per_query_mean']:.0f} (max {st0['candidates_per_query_max']} -> {st1['candidates_per_query_max']})")
    # k larger than the number of chunks a query can rank: the threshold is unusable, the flag answers
    tiny = raglite_amd.DeviceIndex(E[:66_000], np.concatenate((np.arange(0, 66_000, 1000), [66_000])).astype(np.int64), metric="dot")
    s2, c2 = tiny.maxsim_topk_batch(Q[:8], 100)
    with tiny.options(hi_maxsim=0):
        s3, c3 = tiny.maxsim_topk_batch(Q[:8], 100)
    assert torch.equal(c2, c3) and int((c2 >= 0).sum()) == 8 * 66
    assert float((s2[c2 >= 0] - s3[c3 >= 0]).abs().max()) <= 2e-6 * float(s3[c3 >= 0].abs().max())
    tiny.close()
    idx.close()


def test_the_shipped_library_ignores_experiment_environment_variables(monkeypatch):
    """RAGLITE_PP_DBG=2 selects a kernel without MFMAs in the experiments build (wrong results by design).  The shipped library has no
    such kernel and reads no environment variable: same bits with the variable set."""
    import torch

    raglite_amd.set_device(0)
    n, dim, nq, n_queries, k = 70_000, 1024, 32, 16, 20
    rng = np.random.default_rng(92)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=9200)
    Q = torch.empty((n_queries, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=9201)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    a0, _ = idx.maxsim_approx_scores(Q, kernel=0)
    s0, c0 = idx.maxsim_topk_batch(Q, k)
    for name, value in (("RAGLITE_PP_DBG", "2"), ("RAGLITE_GEMM_DBG", "2"), ("RAGLITE_NO_HI_MAXSIM", "1"), ("RAGLITE_HI_ONE_PRODUCT", "0"),
                        ("RAGLITE_NO_PP", "1"), ("RAGLITE_EXACT_FP32", "1")):
        monkeypatch.setenv(name, value)
    idx2 = raglite_amd.DeviceIndex(E, off, metric="dot")  # (an index created with the variables set: nothing is read at creation either)
    assert idx2.arithmetic == "f16_split"
    a1, _ = idx2.maxsim_approx_scores(Q, kernel=0)
    s1, c1 = idx2.maxsim_topk_batch(Q, k)
    assert idx2.filter_stats()["kind"] == "maxsim_batch_hi"
    assert torch.equal(a0, a1) and torch.equal(s0, s1) and torch.equal(c0, c1)
    idx.close()
    idx2.close()


# ---- index footprint: rows + HI image only (RL_OPT_KEEP_IMAGE = 0, RL_OPT_KEEP_HI_PLANE = 0 -- 1.5 x the corpus instead of 3 x) -----------------
# The bound-filtered batch reads the HI image (approximate pass) and the rows (exact re-scoring); only its guarded full-precision fallback
# ever read the pre-split image.  Without that image the fallback runs the streaming kernels over the rows (one launch, grid row = query).


def test_slim_index_rows_plus_hi_image_same_bits():
    torch = _torch()
    rng = np.random.default_rng(41)
    off = ragged_offsets(rng, N, 1, 15)
    E = _corpus(torch, N, DIM, seed=12_000)
    Qb = _queries(torch, 19, 32, DIM, seed=12_001)  # 16 + 3: two passes of the sixteen-query kernel
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    idx.set_option("lazy_images", 0)  # every image at once, as before round 5: the options below release them one by one
    m0 = idx.memory()
    assert m0["presplit_image"] > 0 and m0["hi_plane"] > 0 and m0["hi_image"] > 0
    s0, c0 = idx.maxsim_topk_batch(Qb, 100)
    st0 = idx.filter_stats()
    idx.set_option("keep_image", 0)
    idx.set_option("keep_hi_plane", 0)
    m1 = idx.memory()
    assert m1["presplit_image"] == 0 and m1["hi_plane"] == 0 and m1["hi_image"] == m0["hi_image"]
    assert m1["rows"] + m1["hi_image"] <= 1.51 * m1["rows"]
    s1, c1 = idx.maxsim_topk_batch(Qb, 100)
    st1 = idx.filter_stats()
    assert st1["kind"] == st0["kind"] == "maxsim_batch_hi" and not st1["fallback"] and not st0["fallback"]
    assert st1["candidates_per_query_max"] == st0["candidates_per_query_max"]
    assert torch.equal(c0, c1) and torch.equal(s0, s1)
    # the staged form (what a shard of a ShardedIndex runs) on the slim index
    a = idx.maxsim_batch_begin(Qb, 100)
    s2, c2 = idx.maxsim_batch_finish(Qb, a[None], 0, 100)
    assert torch.equal(c0.to(torch.int64), c2.to(torch.int64)) and torch.equal(s0, s2)
    # the eight-query kernel reads the pre-split image's layout partner: not available here, the call says so instead of reading freed memory
    with idx.options(pp_pass=0):
        s3, c3 = idx.maxsim_topk_batch(Qb, 100)  # (falls to the streaming kernels: same chunks, scores within the paths' rounding)
    assert idx.filter_stats()["kind"] == "none"
    for i in range(Qb.shape[0]):
        assert set(c3[i].tolist()) == set(c0[i].tolist())
    torch.testing.assert_close(s3.sort(dim=1).values, s0.sort(dim=1).values, rtol=0, atol=2e-6 * float(s0.abs().max()))
    # appends keep the slim layout and its results
    extra = _corpus(torch, 4_000, DIM, seed=12_002)
    idx.append(extra, np.full(1000, 4, dtype=np.int64))
    full = raglite_amd.DeviceIndex(torch.cat([E, extra]), np.concatenate([off, off[-1] + 4 * np.arange(1, 1001)]), metric="dot")
    m2 = idx.memory()
    assert m2["presplit_image"] == 0 and m2["hi_plane"] == 0 and m2["hi_image"] > 0
    s4, c4 = idx.maxsim_topk_batch(Qb, 100)
    s5, c5 = full.maxsim_topk_batch(Qb, 100)
    assert torch.equal(c4, c5) and torch.equal(s4, s5)
    idx.close()
    full.close()


@pytest.mark.parametrize("data", ["integer_ties", "near_identical"])
def test_slim_index_fallback_runs_the_streaming_kernels(data):
    """More candidates than a list holds: the flag goes up and the guarded fallback answers -- over the rows when there is no pre-split image."""
    rng = np.random.default_rng(43)
    off = np.arange(N + 1, dtype=np.int64)
    if data == "integer_ties":  # 6 000 identical one-row chunks at the top of every ranking; integer data: every path is exact
        E = oracle.synth_matrix(12_100, N, DIM, "small_int")
        Qb = np.stack([oracle.synth_matrix(12_200 + i, 8, DIM, "small_int") for i in range(5)])
        hot = rng.choice(N, 6000, replace=False)
        E[hot] = np.sign(Qb[:, 0].sum(axis=0))[None, :] * 3.0
    else:
        E = oracle.synth_matrix(12_300, N, DIM)
        Qb = np.stack([oracle.synth_matrix(12_400 + i, 8, DIM) for i in range(5)])
        hot = rng.choice(N, 4000, replace=False)
        E[hot] = (3.0 * Qb[:, 0].sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, DIM))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    fs, fc = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"]
    idx.set_option("keep_image", 0)
    idx.set_option("keep_hi_plane", 0)
    assert idx.memory()["presplit_image"] == 0
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and st["fallback"]
    if data == "integer_ties":
        assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
        for i in (0, 4):
            ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
            order = np.lexsort((np.arange(len(ref)), -ref))[:100]
            assert np.array_equal(bc[i], order) and np.array_equal(bs[i].astype(np.float64), ref[order])
    else:  # float data: the streaming kernels and the eight-query pass sum in different orders -- same chunks, scores within the rounding
        for i in range(5):
            ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
            assert_topk_close(bs[i], bc[i], ref, 100, 2e-6 * float(np.abs(ref).max()))
    idx.close()


def test_streaming_fallback_of_a_big_batch_runs_on_a_narrower_grid_with_the_same_bits():
    """Rows + HI image (lazy images, first batch): 32 queries fall back through the streaming kernels on n_cu / 8 grid columns per query
    (the guarded launch of a 128-query step would otherwise be 32 k workgroups that return at once).  Integer data: exact, whatever the grid."""
    rng = np.random.default_rng(47)
    off = np.arange(N + 1, dtype=np.int64)
    E = oracle.synth_matrix(12_500, N, DIM, "small_int")
    Qb = np.stack([oracle.synth_matrix(12_600 + i, 8, DIM, "small_int") for i in range(32)])
    hot = rng.choice(N, 6000, replace=False)
    E[hot] = np.sign(Qb[:, 0].sum(axis=0) + 0.5)[None, :] * 3.0  # 6 000 identical one-row chunks at the top of every ranking
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and st["fallback"] and idx.memory()["presplit_image"] == 0
    for i in (0, 13, 31):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        order = np.lexsort((np.arange(len(ref)), -ref))[:100]
        assert np.array_equal(bc[i], order) and np.array_equal(bs[i].astype(np.float64), ref[order])
    idx.close()


def test_fewer_live_chunks_than_k_never_returns_a_masked_chunk():
    """Tombstones (or a metadata filter) that leave fewer than k eligible chunks: the approximate top-k then holds MASKED chunks at -inf, whose
    exact scores are perfectly finite -- the second threshold must not count them (round 6, found by scripts/soak_wide.py: they came back as
    results); the guarded full-precision pass, which masks its scores, answers, padded with (-inf, -1)."""
    rng = np.random.default_rng(19)
    off = ragged_offsets(rng, N, 1, 15)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(10_900, N, DIM)
    Qb = np.stack([oracle.synth_matrix(10_950 + i, 32, DIM) for i in range(8)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    keep = rng.choice(n_chunks, 37, replace=False)
    dead = np.setdiff1d(np.arange(n_chunks), keep)
    idx.delete_chunks(dead)
    s, c = idx.maxsim_topk_batch(Qb, 100)
    assert idx.filter_stats()["fallback"]
    for i in (0, 7):
        ref = oracle.maxsim_scores(E, off, Qb[i], np.float64)
        assert np.isin(c[i][:37], keep).all() and len(set(c[i][:37].tolist())) == 37 and (c[i][37:] == -1).all() and np.isneginf(s[i][37:]).all()
        np.testing.assert_allclose(s[i][:37], ref[c[i][:37]], rtol=0, atol=2e-6 * float(np.abs(ref[keep]).max()))
        assert np.all(np.diff(s[i][:37]) <= 0)
    s1, c1 = idx.maxsim_topk(Qb[0], 100)
    assert np.isin(c1[:37], keep).all() and (c1[37:] == -1).all()
    idx.close()
