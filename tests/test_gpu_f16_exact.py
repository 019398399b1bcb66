"""fp16 queries x fp16-stored corpus (`rl_maxsim_topk_batch_f16`): the one-product pass IS the score.

RAGLite's stored embeddings and its query embeddings are fp16 values (`/root/reference/src/raglite/_embed.py:140,164`; the query
adapter's result is cast back to the query's dtype, `_search.py:62`).  The product of two fp16 values is exact in fp32, an fp16-stored
index has no dropped half, an fp16 query has none either: the sixteen-query pass (`maxsim_pp.hip`) accumulates q . e itself and its
exact top-k is returned -- no error bound, no candidate list, no re-scoring kernel.

score[c] = sum_i max_{j in chunk c} Q[i].D[j], the multi-vector generalisation of `_search.py:143-149` behind the reranker plugin call
(`:394-396`).  Bars: integer-valued data bit-identical to the fp32-accumulating oracle (scores and chunk ordinals, ties included) and to
the bound-filtered route (`f16_exact = 0`); float data within 2^-12 relative of float64 with a tie-aware top-k check, and the same
chunks as the bound-filtered route; queries the pass's power-of-two scaling cannot hold exactly, indexes with fewer than k chunks and
tombstones take the guarded fallback / the masks and still agree."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, ragged_offsets

pytestmark = pytest.mark.gpu

N, DIM = 70_000, 1024  # >= 64 M elements: the index keeps the image the sixteen-query pass reads


def _stats(idx):
    st = idx.filter_stats()
    return st["kind"], st["candidates_per_query_max"], bool(st["fallback"])


@pytest.mark.parametrize("nq,n_queries,k,dim", [(32, 19, 100, 1024), (17, 16, 50, 1024), (1, 3, 10, 1024), (32, 8, 100, 256)])
def test_integer_data_bit_exact_and_route_taken(nq, n_queries, k, dim):
    rng = np.random.default_rng(nq * 100 + n_queries)
    n = N * 1024 // dim
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(20_000 + nq, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(20_100 + i, nq, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E.astype(np.float16), off, metric="dot", storage="f16")
    bs, bc = idx.maxsim_topk_batch(Qb.astype(np.float16), k)
    assert _stats(idx) == ("maxsim_batch_f16_exact", 0, False)
    with idx.options(f16_exact=0):
        fs, fc = idx.maxsim_topk_batch(Qb.astype(np.float16), k)
    assert _stats(idx)[0] == "maxsim_batch_hi"
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    gs, gc = idx.maxsim_topk_batch(Qb, k)  # the same values handed over as fp32: the bound-filtered route, the same bits
    assert np.array_equal(bc, gc) and np.array_equal(bs.view(np.uint32), gs.view(np.uint32))
    for i in sorted({0, n_queries // 2, n_queries - 1}):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i], ws)
    idx.close()


def test_float_data_device_tensors_tolerance_and_agreement_with_the_filtered_route():
    import torch

    rng = np.random.default_rng(9)
    off = ragged_offsets(rng, N, 1, 15)
    E16 = oracle.synth_matrix(20_200, N, DIM).astype(np.float16)
    Q16 = np.stack([oracle.synth_matrix(20_300 + i, 32, DIM) for i in range(21)]).astype(np.float16)
    idx = raglite_amd.DeviceIndex(torch.as_tensor(E16, device="cuda"), off, metric="dot", storage="f16")
    k = 100
    Qd = torch.as_tensor(Q16, device="cuda")
    bs, bc = idx.maxsim_topk_batch(Qd, k)
    assert _stats(idx) == ("maxsim_batch_f16_exact", 0, False)
    with idx.options(f16_exact=0):
        fs, fc = idx.maxsim_topk_batch(Qd, k)
    bs, bc, fs, fc = (x.cpu().numpy() for x in (bs, bc, fs, fc))
    E64 = E16.astype(np.float64)
    for i in range(21):
        ref = oracle.maxsim_scores(E64, off, Q16[i].astype(np.float64), np.float64)
        tol = 2.0 ** -12 * float(np.abs(ref).max())  # the bar of the route: 2^-12 relative
        assert_topk_close(bs[i], bc[i], ref, k, tol)
        # measured: the fp32 sums of 1024 exact products are ~1e-6 relative of float64 -- hold the route to 2e-6 of the score scale too
        np.testing.assert_allclose(bs[i], ref[bc[i]], rtol=0, atol=2e-6 * float(np.abs(ref).max()))
        assert set(bc[i].tolist()) == set(fc[i].tolist())  # (the two routes' scores differ in the last bits: order may too)
        np.testing.assert_allclose(np.sort(bs[i]), np.sort(fs[i]), rtol=0, atol=4e-6 * float(np.abs(ref).max()))
    # a contiguous slice of the caller's fp16 tensor that starts at an element offset not divisible by four (2-byte aligned only):
    # round 5 answered "must be 8-byte aligned"; the values are what matters
    flat = torch.zeros(Qd.numel() + 8, dtype=torch.float16, device="cuda")
    for shift in (1, 2, 3):
        view = flat[shift : shift + Qd.numel()].view(Qd.shape)
        view.copy_(Qd)
        assert view.is_contiguous() and view.data_ptr() % 8 == 2 * shift
        us, uc = idx.maxsim_topk_batch(view, k)
        assert _stats(idx) == ("maxsim_batch_f16_exact", 0, False)
        assert np.array_equal(us.cpu().numpy(), bs) and np.array_equal(uc.cpu().numpy(), bc)
    idx.close()


def test_query_that_loses_a_bit_under_scaling_takes_the_guarded_fallback():
    """One query holds 60 000 next to 2^-24 (fp16's smallest subnormal): scaled so that its largest element sits in [2^13, 2^14) the small
    one is no fp16 value any more -- the flag goes up and the full-precision passes answer the batch, with the filtered route's results."""
    rng = np.random.default_rng(10)
    off = ragged_offsets(rng, N, 1, 15)
    E16 = oracle.synth_matrix(20_400, N, DIM, "small_int").astype(np.float16)
    Q16 = np.stack([oracle.synth_matrix(20_500 + i, 8, DIM, "small_int") for i in range(5)]).astype(np.float16)
    Q16[3, 2, 7] = np.float16(60000.0)
    Q16[3, 2, 8] = np.float16(2.0 ** -24)
    idx = raglite_amd.DeviceIndex(E16, off, metric="dot", storage="f16")
    bs, bc = idx.maxsim_topk_batch(Q16, 20)
    assert _stats(idx) == ("maxsim_batch_f16_exact", 0, True)
    with idx.options(f16_exact=0):
        fs, fc = idx.maxsim_topk_batch(Q16, 20)
    assert np.array_equal(bc, fc)
    np.testing.assert_allclose(bs, fs, rtol=1e-6, atol=0)
    for i in (0, 3):
        ref = oracle.maxsim_scores(E16.astype(np.float64), off, Q16[i].astype(np.float64), np.float64)
        assert_topk_close(bs[i], bc[i], ref, 20, 2e-6 * float(np.abs(ref).max()))
    idx.close()


def test_fewer_chunks_than_k_tombstones_and_small_batches():
    rng = np.random.default_rng(11)
    n = 66_000
    off = np.concatenate((np.arange(0, n, 1100), [n])).astype(np.int64)  # 60 chunks of 1100 rows
    E16 = oracle.synth_matrix(20_600, n, DIM, "small_int").astype(np.float16)
    Q16 = np.stack([oracle.synth_matrix(20_700 + i, 16, DIM, "small_int") for i in range(6)]).astype(np.float16)
    idx = raglite_amd.DeviceIndex(E16, off, metric="dot", storage="f16")
    k = 100  # > 60 chunks: padding (-inf, -1) behind the 60 real entries
    bs, bc = idx.maxsim_topk_batch(Q16, k)
    with idx.options(f16_exact=0):
        fs, fc = idx.maxsim_topk_batch(Q16, k)
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    assert (bc[:, 60:] == -1).all() and np.isneginf(bs[:, 60:]).all() and (bc[:, :60] >= 0).all()
    # tombstones: deleted chunks never come back, the rest ranks as before
    dead = np.unique(bc[:, :3])
    idx.delete_chunks(dead)
    bs2, bc2 = idx.maxsim_topk_batch(Q16, 20)
    assert not np.isin(bc2, dead).any()
    with idx.options(f16_exact=0):
        fs2, fc2 = idx.maxsim_topk_batch(Q16, 20)
    assert np.array_equal(bc2, fc2) and np.array_equal(bs2.view(np.uint32), fs2.view(np.uint32))
    for i in (0, 5):
        ws = oracle.maxsim_scores(E16.astype(np.float32), off, Q16[i].astype(np.float32), np.float32).copy()
        ws[dead] = -np.inf
        es, ec = oracle.topk_desc(ws, 20)
        assert np.array_equal(bc2[i], ec) and np.array_equal(bs2[i], es.astype(np.float32))
    # one or two queries: no shared pass, the streaming kernels -- fp16 queries are just widened
    s1, c1 = idx.maxsim_topk_batch(Q16[:2], 10)
    for i in range(2):
        ws = oracle.maxsim_scores(E16.astype(np.float32), off, Q16[i].astype(np.float32), np.float32).copy()
        ws[dead] = -np.inf
        es, ec = oracle.topk_desc(ws, 10)
        assert np.array_equal(c1[i], ec) and np.array_equal(s1[i], es.astype(np.float32))
    idx.close()


def test_fp16_queries_over_an_fp32_index_of_fp16_values_take_the_exact_route_too():
    """RAGLite's embeddings handed over as float32 arrays: the index measures what its HI halves drop when it builds their image -- exactly
    zero here -- and the fp16 queries' pass is exact again."""
    rng = np.random.default_rng(12)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(20_800, N, DIM, "small_int")
    Q16 = np.stack([oracle.synth_matrix(20_900 + i, 32, DIM, "small_int") for i in range(8)]).astype(np.float16)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Q16, 100)
    assert _stats(idx) == ("maxsim_batch_f16_exact", 0, False)
    with idx.options(f16_exact=0):
        fs, fc = idx.maxsim_topk_batch(Q16, 100)
    assert _stats(idx)[0] == "maxsim_batch_hi"
    assert np.array_equal(bc, fc) and np.array_equal(bs.view(np.uint32), fs.view(np.uint32))
    for i in (0, 7):
        ws, wc = oracle.maxsim_topk(E, off, Q16[i].astype(np.float32), 100, np.float32)
        assert np.array_equal(bc[i], wc) and np.array_equal(bs[i], ws)
    # unit rows rounded through fp16 (`_embed.py:139-140`), stored as float32: float data, the same route, the route's tolerance
    U = oracle.synth_matrix(20_850, N, DIM)
    U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    Qu = np.stack([oracle.synth_matrix(20_950 + i, 32, DIM) for i in range(5)])
    Qu = (Qu / np.linalg.norm(Qu, axis=2, keepdims=True)).astype(np.float16)
    idu = raglite_amd.DeviceIndex(U, off, metric="dot")
    us, uc = idu.maxsim_topk_batch(Qu, 100)
    assert _stats(idu) == ("maxsim_batch_f16_exact", 0, False)
    for i in range(5):
        ref = oracle.maxsim_scores(U.astype(np.float64), off, Qu[i].astype(np.float64), np.float64)
        assert_topk_close(us[i], uc[i], ref, 100, 1e-4)  # north_star's absolute bar on unit-norm scores (<= 32)
        np.testing.assert_allclose(us[i], ref[uc[i]], rtol=0, atol=2e-5)
    idu.close()
    idx.close()


def test_fp16_queries_over_an_fp32_index_of_other_values_are_widened_and_filtered_as_before():
    rng = np.random.default_rng(13)
    off = ragged_offsets(rng, N, 1, 15)
    E = oracle.synth_matrix(20_860, N, DIM)  # U(-1, 1): not fp16 values
    Q16 = np.stack([oracle.synth_matrix(20_960 + i, 32, DIM) for i in range(8)]).astype(np.float16)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Q16, 100)
    assert _stats(idx)[0] == "maxsim_batch_hi"
    for i in (0, 7):
        ref = oracle.maxsim_scores(E, off, Q16[i].astype(np.float64), np.float64)
        assert_topk_close(bs[i], bc[i], ref, 100, 2e-6 * float(np.abs(ref).max()))
    idx.close()
