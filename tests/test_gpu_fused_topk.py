"""GPU parity of the fused exact batched row top-k (api.hip `search_rows_fused`, maxsim_gemm.hip MODE 2): batches of
>= 96 queries over >= 4096 rows rank WITHOUT the [B x N] score matrix -- a sampled GEMM pass gives every query a lower
bound of its k-th best score, a second pass keeps only the rows that reach it, and the merge kernel ranks those lists.

The contract is "same bits as the dense GEMM + selection path" (the reference ranks every row,
/root/reference/src/raglite/_search.py:75-79 `ORDER BY ... LIMIT`): checked against the oracle on integer data, against
the dense path (RAGLITE_NO_FUSED_TOPK=1) on float data, and with the candidate lists forced to overflow
(RAGLITE_FUSED_TOPK_CAP) so that the guarded dense fallback is the one that answers.
"""

import os

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, sim_fp32_exact

pytestmark = pytest.mark.gpu
TOL = 2e-6  # relative to the score scale (|e| |q| for dot products), against float64


def _tol(E, q, metric):
    if metric == "cosine":
        return TOL
    return TOL * max(1.0, float(np.linalg.norm(E, axis=1).max() * np.linalg.norm(q)))



@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("n,dim,B,k", [(4096, 64, 96, 10), (5000, 128, 128, 100), (20_001, 384, 200, 7), (70_000, 1024, 97, 512),
                                        (33_333, 256, 1000, 1)])
def test_fused_equals_dense_path_bitwise(metric, n, dim, B, k):
    E = oracle.synth_matrix(7000 + n, n, dim)
    Q = oracle.synth_matrix(7100 + B, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    with idx.options(fused_hi=0):  # (the 70 000 x 1024 case keeps a HI image: its default is the fused top-k over THAT, tested below)
        S, R = idx.search_rows(Q, k)
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Q, k)
    assert np.array_equal(R, R0)
    assert np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    for b in (0, B // 2, B - 1):
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], metric), k, _tol(E, Q[b], metric))
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_fused_integer_ties_bit_exact(metric):
    """Small-integer data: thousands of rows share every score, the bound itself is a tie value -- ties resolve to the lowest
    row exactly as the oracle's stable sort does, whether the lists hold them or the fallback answers."""
    n, dim, B, k = 30_000, 64, 130, 64
    E = oracle.synth_matrix(7200, n, dim, "small_int")
    Q = oracle.synth_matrix(7201, B, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    for b in (0, 1, 64, 129):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei)
        assert np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Q, k)
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    idx.close()


def test_fused_constant_corpus_overflows_into_fallback():
    """Every row identical: each query's list would need all N rows -> overflow flag -> the guarded dense pass; the answer is
    rows 0..k-1 with one score."""
    n, dim, B, k = 10_000, 128, 100, 20
    E = np.tile(oracle.synth_matrix(7300, 1, dim), (n, 1))
    Q = oracle.synth_matrix(7301, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    S, R = idx.search_rows(Q, k)
    assert np.array_equal(R, np.tile(np.arange(k, dtype=R.dtype), (B, 1)))
    assert (S == S[:, :1]).all()
    idx.close()


@pytest.mark.parametrize("cap", ["16", "64", "700"])
def test_forced_list_overflow_falls_back_exactly(cap):
    n, dim, B, k = 50_000, 128, 128, 10
    E = oracle.synth_matrix(7400, n, dim)
    Q = oracle.synth_matrix(7401, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Q, k)
    with idx.options(fused_topk_cap=int(cap)):
        S, R = idx.search_rows(Q, k)
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    S, R = idx.search_rows(Q, k)  # and the flag is re-armed per call: the next batch takes the lists again
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    idx.close()


def test_fused_after_append_and_delete():
    """The image follows appends; deletions put a row mask on the search, which routes to the dense path -- both agree with
    a fresh index over the same rows."""
    n, dim, B, k = 12_000, 128, 100, 25
    E = oracle.synth_matrix(7500, n, dim)
    Q = oracle.synth_matrix(7501, B, dim)
    idx = raglite_amd.DeviceIndex(E[:8000], metric="cosine")
    idx.append(E[8000:])
    S, R = idx.search_rows(Q, k)
    ref = raglite_amd.DeviceIndex(E, metric="cosine")
    with ref.options(fused_topk=0):
        S0, R0 = ref.search_rows(Q, k)
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    for b in (0, 50, 99):
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], "cosine"), k, TOL)
    dead = np.unique(R[:, 0])[:40]
    idx.delete_chunks(dead)
    S1, R1 = idx.search_rows(Q, k)
    assert not np.isin(R1, dead).any()
    for b in (0, 99):
        sim = oracle.similarity(E, Q[b], "cosine").copy()
        sim[dead] = -np.inf
        assert_topk_close(S1[b], R1[b], sim, k, TOL)
    idx.compact()  # tombstones squeezed out: no mask any more, the lists answer again
    live = np.setdiff1d(np.arange(n), dead)
    S2, R2 = idx.search_rows(Q, k)
    assert np.array_equal(live[R2], R1) and np.array_equal(S2.view(np.uint32), S1.view(np.uint32))
    idx.close()
    ref.close()


@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_f16_storage_big_batch_over_the_image(metric):
    """fp16-stored index, B >= 96: the row-score GEMM (and, cosine / dot, the fused top-k) over the one-plane image.  Integer
    data is exact in fp16 -> bit-identical to the fp32 oracle; unit-norm fp16 rows -> tolerance against float64."""
    n, dim, B, k = 9000, 128, 130, 30
    E = oracle.synth_matrix(7600, n, dim, "small_int")
    Q = oracle.synth_matrix(7601, B, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric, storage="f16")
    S, R = idx.search_rows(Q, k)
    for b in (0, 64, B - 1):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei)
        assert np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    idx.close()
    Ef = oracle.synth_matrix(7602, n, dim)
    E16 = (Ef / np.linalg.norm(Ef, axis=1, keepdims=True)).astype(np.float16)
    Qf = oracle.synth_matrix(7603, B, dim)
    idx = raglite_amd.DeviceIndex(E16, metric=metric, storage="f16")
    S, R = idx.search_rows(Qf, k)
    Ev = E16.astype(np.float32)
    for b in (0, B - 1):
        assert_topk_close(S[b], R[b], oracle.similarity(Ev, Qf[b], metric), k, 1e-5 if metric == "l2" else _tol(Ev, Qf[b], metric))
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Qf, k)
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    idx.close()


# ---- the fused top-k over the HI image at one MFMA product per multiply (api.hip search_rows_fused_hi): the default for big batches over
# an fp32 corpus of >= 64 M elements (it keeps a HI image) since round 3; RAGLITE_NO_FUSED_HI=1 restores the fused top-k over the
# pre-split image, RAGLITE_NO_FUSED_TOPK=1 the dense path. ------------------------------------------------------------------------------


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_fused_hi_float_data(metric):
    n, dim, B, k = 70_000, 1024, 130, 100
    E = oracle.synth_matrix(7300, n, dim)
    Q = oracle.synth_matrix(7301, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["kind"] == "rows_fused_hi" and not idx.filter_stats()["fallback"]
    with idx.options(fused_hi=0):
        S0, R0 = idx.search_rows(Q, k)  # the fused top-k over the pre-split image (three products)
    assert idx.filter_stats()["kind"] == "rows_fused"
    for b in range(0, B, 13):
        tol = _tol(E, Q[b], metric)
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], metric), k, tol)
        assert len(set(R[b].tolist()) ^ set(R0[b].tolist())) <= 2  # (scores within an ulp of the k-th may swap at the boundary)
        np.testing.assert_allclose(S[b], S0[b], rtol=0, atol=2 * tol)
    idx.close()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_fused_hi_integer_ties_bit_exact(metric):
    n, dim, B, k = 70_000, 1024, 100, 64
    E = oracle.synth_matrix(7400, n, dim, "small_int")
    Q = oracle.synth_matrix(7401, B, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    S, R = idx.search_rows(Q, k)
    for b in (0, 1, 50, 99):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], metric), k)
        assert np.array_equal(R[b], ei)
        if metric == "dot":  # (cosine: the exact re-scoring divides the same integers by the same norms, but sums in another order than NumPy)
            assert np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    idx.close()


def test_fused_hi_near_duplicates_fall_back():
    """3 000 rows within 1e-5 of each other at the top of every ranking: more rows inside the error band than a re-scoring list
    holds -> the device flag -> the dense full-precision path answers, bit for bit what it answers without the switch."""
    rng = np.random.default_rng(75)
    n, dim, B, k = 70_000, 1024, 100, 50
    E = oracle.synth_matrix(7500, n, dim)
    Q = oracle.synth_matrix(7501, B, dim)
    hot = rng.choice(n, 3000, replace=False)
    E[hot] = (Q.sum(axis=0)[None, :] + 1e-5 * rng.standard_normal((3000, dim))).astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["kind"] == "rows_fused_hi" and idx.filter_stats()["fallback"]
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Q, k)
    assert np.array_equal(R, R0) and np.array_equal(S.view(np.uint32), S0.view(np.uint32))
    idx.close()



# ---- round 4: the candidate pass of the fused top-k over the HI image on the sixteen-group tile of maxsim_pp.hip (MODE 2; option fused_pp) ----
@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("n,dim,B,k", [
    (70_000, 1024, 130, 100),    # one partial query tile (5 groups of 16)
    (70_001, 1024, 1000, 10),    # two query tiles (the cfg 5 batch), n not a multiple of 128 / 16
    (66_000, 1024, 2000, 7),     # four query tiles over 516 row tiles: workgroups whose pairs cross query tiles
    (140_000, 512, 600, 100),    # 16 K slabs per tile
    (262_200, 256, 513, 33),     # 8 K slabs per tile (the smallest the tile takes); a query tile with ONE query
    (66_000, 1024, 96, 512),     # k = 512: sample stride at its lower end
])
def test_fused_pp_tile_equals_the_eight_group_tile_bitwise(metric, n, dim, B, k):
    """Both candidate passes multiply the same fp16 halves and keep the rows whose approximate similarity reaches the same threshold;
    what is returned are the exact fp32 similarities of the same candidates: the same bits, whichever tile found them."""
    import torch

    raglite_amd.set_device(0)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=8000 + n % 97)
    Q = torch.empty((B, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=8100 + B)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    assert idx.get_option("fused_pp") == 1
    S, R = idx.search_rows(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "rows_fused_hi" and not st["fallback"], st
    with idx.options(fused_pp=0):
        S0, R0 = idx.search_rows(Q, k)
        st0 = idx.filter_stats()
    assert st0["kind"] == "rows_fused_hi" and not st0["fallback"]
    assert torch.equal(R, R0) and torch.equal(S.view(torch.int32), S0.view(torch.int32))
    # fewer candidates: the sixteen-group tile runs in two rounds and tightens its thresholds after the first (and keeps exactly the rows
    # that reach the threshold; the eight-group tile also those within the 1e-5 of slack of its in-loop test); in ONE round about the same
    assert st["candidates_per_query_mean"] <= st0["candidates_per_query_mean"], (st, st0)
    with idx.options(fused_two_rounds=0):
        S1, R1 = idx.search_rows(Q, k)
        st1 = idx.filter_stats()
    assert torch.equal(R, R1) and torch.equal(S.view(torch.int32), S1.view(torch.int32))
    assert st1["candidates_per_query_mean"] <= st0["candidates_per_query_mean"] <= 1.05 * st1["candidates_per_query_mean"] + 1, (st1, st0)
    # round 5: the SAMPLE pass on the sixteen-group tile too (MODE 1; option fused_pp_sample) -- the thresholds it leads to agree with the
    # eight-group kernel's to the last bits of an fp32 sum, the candidates they admit to within a few rows, the results bit for bit
    assert idx.get_option("fused_pp_sample") == 1
    with idx.options(fused_pp_sample=0):
        S2, R2 = idx.search_rows(Q, k)
        st2 = idx.filter_stats()
    assert st2["kind"] == "rows_fused_hi" and not st2["fallback"]
    assert torch.equal(R, R2) and torch.equal(S.view(torch.int32), S2.view(torch.int32))
    assert abs(st["candidates_per_query_mean"] - st2["candidates_per_query_mean"]) <= 0.02 * st2["candidates_per_query_mean"] + 1, (st, st2)
    # ... and the lists cut by a radix select of their k-th best score (option list_select) instead of by sorting them: the same k-th score,
    # hence the same raised thresholds, the same rows to re-score, the same bits
    assert idx.get_option("list_select") == 1
    with idx.options(list_select=0):
        S3, R3 = idx.search_rows(Q, k)
        st3 = idx.filter_stats()
    assert torch.equal(R, R3) and torch.equal(S.view(torch.int32), S3.view(torch.int32))
    assert st3["candidates_per_query_mean"] == st["candidates_per_query_mean"] and st3["candidates_per_query_max"] == st["candidates_per_query_max"]
    Eh = E.cpu().numpy()
    for b in (0, B // 2, B - 1):
        assert_topk_close(S[b].cpu().numpy(), R[b].cpu().numpy(), oracle.similarity(Eh, Q[b].cpu().numpy(), metric), k, _tol(Eh, Q[b].cpu().numpy(), metric))
    idx.close()


@pytest.mark.parametrize("cap", [16, 64])
def test_fused_pp_forced_overflow_falls_back_exactly(cap):
    n, dim, B, k = 70_000, 1024, 200, 10
    E = oracle.synth_matrix(8300, n, dim)
    Q = oracle.synth_matrix(8301, B, dim)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    with idx.options(fused_topk=0):
        S0, R0 = idx.search_rows(Q, k)
    with idx.options(fused_topk_cap=cap, fused_hi=0):  # (the capacity switch acts on the lists of the pre-split-image path)
        S1, R1 = idx.search_rows(Q, k)
        assert idx.filter_stats()["fallback"]
    assert np.array_equal(R1, R0) and np.array_equal(S1.view(np.uint32), S0.view(np.uint32))
    S, R = idx.search_rows(Q, k)  # the sixteen-group tile again, flag re-armed
    assert idx.filter_stats()["kind"] == "rows_fused_hi" and not idx.filter_stats()["fallback"]
    for b in (0, 100, 199):
        assert_topk_close(S[b], R[b], oracle.similarity(E, Q[b], "cosine"), k, TOL)
    idx.close()


def test_fused_pp_integer_ties_and_lifecycle():
    """Thousands of ties on every threshold (integer data), then append / delete / compact under the sixteen-group tile."""
    n, dim, B, k = 70_000, 1024, 160, 64
    E = oracle.synth_matrix(8400, n + 4000, dim, "small_int")
    Q = oracle.synth_matrix(8401, B, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E[:n], metric="dot")
    idx.append(E[n:])
    S, R = idx.search_rows(Q, k)
    assert idx.filter_stats()["kind"] == "rows_fused_hi"
    for b in (0, 77, 159):
        es, ei = oracle.topk_desc(sim_fp32_exact(E, Q[b], "dot"), k)
        assert np.array_equal(R[b], ei) and np.array_equal(S[b].view(np.uint32), es.astype(np.float32).view(np.uint32))
    dead = np.unique(R[:, 0])[:30]
    idx.delete_chunks(dead)
    S1, R1 = idx.search_rows(Q, k)
    assert not np.isin(R1, dead).any()
    idx.compact()
    S2, R2 = idx.search_rows(Q, k)
    live = np.setdiff1d(np.arange(n + 4000), dead)
    assert np.array_equal(live[R2], R1) and np.array_equal(S2.view(np.uint32), S1.view(np.uint32))
    idx.close()


def test_fused_pp_rows_of_wildly_different_norms_stay_on_the_candidate_pass():
    """Cosines over a corpus where a few thousand short rows (|e| ~ 0.02) sit among long ones (|e| ~ 13): the block-wide bound of the
    candidate pass would pass most of every block that holds a short row -- the record logs fill and the dense path answers.  The index
    knows the spread of its row norms (min / max |e|) and launches the variant that tests hits row by row: no fallback, the same result as
    the dense path's ranking, the oracle's rows."""
    import torch

    raglite_amd.set_device(0)
    n, dim, B, k = 162_724, 512, 512, 100
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=8500)
    Q = torch.empty((B, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=8501)
    g = torch.Generator(device="cuda").manual_seed(3)
    hot = torch.randperm(n, device="cuda", generator=g)[:3000]
    E[hot] = Q.mean(dim=0, keepdim=True) + 1e-3 * torch.randn((3000, dim), device="cuda", generator=g)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    S, R = idx.search_rows(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "rows_fused_hi" and not st["fallback"], st
    Eh = E.cpu().numpy()
    for b in (0, B // 2, B - 1):
        qh = Q[b].cpu().numpy()
        assert_topk_close(S[b].cpu().numpy(), R[b].cpu().numpy(), oracle.similarity(Eh, qh, "cosine"), k, _tol(Eh, qh, "cosine"))
    with idx.options(fused_pp=0):  # the eight-group tile tests row by row as well: same candidates' exact similarities, same bits
        S0, R0 = idx.search_rows(Q, k)
        assert not idx.filter_stats()["fallback"]
    assert torch.equal(R, R0) and torch.equal(S.view(torch.int32), S0.view(torch.int32))
    idx.close()
