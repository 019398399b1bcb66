"""GPU parity of the eight-queries-per-pass MaxSim kernel over the pre-split corpus image
(raglite_amd/csrc/maxsim_gemm.hip), through `rl_maxsim_topk_batch` of the C ABI.

score[c] = sum_i max_{j in chunk c} Q[i].D[j] -- the multi-query-vector generalisation of
/root/reference/src/raglite/_search.py:143-149 (per-chunk max) behind the reranker plugin call (:394-396).
Bars: integer-valued data bit-identical to the NumPy oracle (scores AND chunk ordinals, ties included), U(-1,1)
data within 1e-4 relative to the score scale of the float64 oracle with a tie-aware top-k check.
"""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import assert_topk_close, ragged_offsets

pytestmark = pytest.mark.gpu


def _layout(kind: str, rng, n_rows: int) -> np.ndarray:
    if kind == "ragged":
        return ragged_offsets(rng, n_rows, 1, 15)
    if kind == "rows":  # every row its own chunk
        return np.arange(n_rows + 1, dtype=np.int64)
    if kind == "one":  # one chunk spanning every tile of every workgroup
        return np.array([0, n_rows], dtype=np.int64)
    if kind == "long":  # chunks longer than a 256-row tile next to single rows
        sizes, total = [], 0
        while total < n_rows:
            s = int(rng.choice([1, 2, 17, 300, 700]))
            s = min(s, n_rows - total)
            sizes.append(s)
            total += s
        return np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    raise AssertionError(kind)


@pytest.mark.parametrize("dim", [64, 128, 384, 1024])
@pytest.mark.parametrize("n_rows,nq,n_queries,kind", [
    (20, 32, 3, "ragged"), (700, 17, 8, "ragged"), (9000, 32, 11, "ragged"), (40_000, 25, 19, "ragged"),
    (5000, 1, 8, "rows"), (3000, 16, 9, "one"), (20_000, 32, 8, "long"), (257, 5, 4, "rows"), (4097, 32, 16, "ragged"),
])
def test_gemm_pass_integer_bit_exact(dim, n_rows, nq, n_queries, kind):
    rng = np.random.default_rng(dim * 7 + n_rows + nq)
    off = _layout(kind, rng, n_rows)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(1200 + dim, n_rows, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(1300 + i, nq, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    assert idx.arithmetic == "f16_split"
    k = min(50, n_chunks)
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in range(n_queries):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i][: len(wc)], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i][: len(wc)], ws), (i, bs[i][:8], ws[:8])
    idx.close()


def test_gemm_pass_equals_single_query_kernel_on_all_chunks(torch_cuda):
    """Every chunk score (not only the top-k) of a batch equals the single-query kernel's on integer data, for a
    corpus large enough that every workgroup walks several tiles, and the batch composition does not matter."""
    torch = torch_cuda
    n, dim, nq = 70_000, 1024, 32
    rng = np.random.default_rng(5)
    off = ragged_offsets(rng, n, 1, 15)
    n_chunks = len(off) - 1
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=71, kind="small_int")
    Qb = torch.empty((8, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Qb, seed=72, kind="small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, min(2048, n_chunks))
    for i in (0, 3, 7):
        ss, sc = idx.maxsim_topk(Qb[i], min(2048, n_chunks))
        assert torch.equal(bc[i], sc) and torch.equal(bs[i], ss)
    # same queries in another batch composition (5 = one pass of five queries): identical bits
    b5s, b5c = idx.maxsim_topk_batch(Qb[3:8], 100)
    assert torch.equal(b5c, bc[3:8, :100]) and torch.equal(b5s, bs[3:8, :100])
    idx.close()


def test_gemm_pass_uniform_data_tolerance(torch_cuda):
    """U(-1,1) data: the batch kernel sums all of K in one accumulator chain (the streaming kernels sum K quarters), so
    it is not bit-identical to them; both must sit within the fp32 bar of the float64 oracle.  Scores have magnitude
    ~ nq * sqrt(dim / 3) * 3 ~ 600; the bar is 1e-4 of that (north star: 1e-4 on unit-norm rows)."""
    torch = torch_cuda
    n, dim, nq = 30_000, 1024, 32
    rng = np.random.default_rng(77)
    off = ragged_offsets(rng, n, 1, 15)
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=31)
    Qb = torch.empty((9, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Qb, seed=32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    Eh = E.cpu().numpy().astype(np.float64)
    for i in (0, 4, 8):
        S = Eh @ Qb[i].cpu().numpy().astype(np.float64).T  # noqa: N806
        all_scores = np.maximum.reduceat(S, off[:-1], axis=0).sum(axis=1)
        scale = np.abs(all_scores).max()
        assert_topk_close(bs[i].cpu().numpy(), bc[i].cpu().numpy(), all_scores, 100, 2e-6 * scale)
    idx.close()


def test_gemm_pass_unit_norm_rows_within_1e4():
    """The north star's own bar: unit-norm rows (what RAGLite stores), scores within 1e-4 absolute of float64."""
    rng = np.random.default_rng(9)
    n, dim, nq = 12_000, 1024, 32
    E = rng.standard_normal((n, dim)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    Qb = rng.standard_normal((8, nq, dim)).astype(np.float32)
    Qb /= np.linalg.norm(Qb, axis=2, keepdims=True)
    off = ragged_offsets(rng, n, 1, 9)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    assert idx.arithmetic == "f16_split"
    bs, bc = idx.maxsim_topk_batch(Qb, 100)
    E64 = E.astype(np.float64)
    for i in range(8):
        S = E64 @ Qb[i].astype(np.float64).T  # noqa: N806
        all_scores = np.maximum.reduceat(S, off[:-1], axis=0).sum(axis=1)
        assert_topk_close(bs[i], bc[i], all_scores, 100, 1e-4 / 4)  # observed ~2e-6; the bar leaves 4x head room
    idx.close()


def test_gemm_pass_follows_append_delete_and_arithmetic_switch():
    """The corpus image is extended on append (incl. a partial 16-row block), tombstones mask its scores, the exact-fp32
    switch drops it and AUTO rebuilds it: a grown + thinned index equals a fresh one over the same rows, bit for bit."""
    rng = np.random.default_rng(21)
    dim, nq = 256, 32
    E = oracle.synth_matrix(1400, 6000, dim, "small_int")
    off = ragged_offsets(rng, 6000, 1, 15)
    cut_chunk = int(np.searchsorted(off, 3333))
    cut = int(off[cut_chunk])  # not a multiple of 16 in general
    Qb = np.stack([oracle.synth_matrix(1500 + i, nq, dim, "small_int") for i in range(8)])
    idx = raglite_amd.DeviceIndex(E[:cut], off[: cut_chunk + 1], metric="dot")
    idx.append(E[cut:], np.diff(off[cut_chunk:]))
    fresh = raglite_amd.DeviceIndex(E, off, metric="dot")
    k = 64
    a_s, a_c = idx.maxsim_topk_batch(Qb, k)
    f_s, f_c = fresh.maxsim_topk_batch(Qb, k)
    assert np.array_equal(a_c, f_c) and np.array_equal(a_s, f_s)
    for i in range(8):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(a_c[i], wc) and np.array_equal(a_s[i], ws)
    dead = rng.choice(len(off) - 1, size=200, replace=False)
    idx.delete_chunks(dead)
    d_s, d_c = idx.maxsim_topk_batch(Qb, k)
    assert not np.isin(d_c, dead).any()
    for i in range(8):
        ss, sc = idx.maxsim_topk(Qb[i], k)
        assert np.array_equal(d_c[i], sc) and np.array_equal(d_s[i], ss)
    idx.set_exact_fp32(True)
    assert idx.arithmetic == "fp32_exact"
    e_s, e_c = idx.maxsim_topk_batch(Qb, k)
    idx.set_exact_fp32(False)
    assert idx.arithmetic == "f16_split"
    g_s, g_c = idx.maxsim_topk_batch(Qb, k)
    assert np.array_equal(e_c, d_c) and np.array_equal(e_s, d_s)  # integer data: every arithmetic is exact
    assert np.array_equal(g_c, d_c) and np.array_equal(g_s, d_s)
    idx.close()
    fresh.close()


# ---- fp16-stored corpus (the reference's pgvector halfvec column, src/raglite/_typing.py:211-232): the same pass over the
# one-plane image, two MFMA products per multiply ------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [128, 384, 1024])  # (an fp16-stored index takes dims 128 ... 1024)
@pytest.mark.parametrize("n_rows,nq,n_queries,kind", [
    (20, 32, 3, "ragged"), (700, 17, 8, "ragged"), (9000, 32, 11, "ragged"), (5000, 1, 8, "rows"), (20_000, 32, 8, "long"),
    (4097, 32, 16, "ragged"),
])
def test_gemm_pass_f16_storage_integer_bit_exact(dim, n_rows, nq, n_queries, kind):
    rng = np.random.default_rng(dim * 11 + n_rows + nq)
    off = _layout(kind, rng, n_rows)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(1400 + dim, n_rows, dim, "small_int")  # exact in fp16
    Qb = np.stack([oracle.synth_matrix(1500 + i, nq, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage="f16")
    assert idx.arithmetic == "f16_stored"
    k = min(50, n_chunks)
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in range(n_queries):
        ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
        assert np.array_equal(bc[i][: len(wc)], wc), (i, bc[i][:8], wc[:8])
        assert np.array_equal(bs[i][: len(wc)], ws), (i, bs[i][:8], ws[:8])
    idx.close()


def test_gemm_pass_f16_storage_float_data_and_lifecycle():
    """Unit-norm rows rounded to fp16 (what RAGLite stores): scores within 1e-5 of the float64 oracle over the STORED values;
    append extends the image, deleted chunks never appear."""
    rng = np.random.default_rng(77)
    n_rows, dim, nq, n_queries = 30_000, 128, 32, 9
    off = ragged_offsets(rng, n_rows, 1, 12)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(1600, n_rows, dim)
    E16 = (E / np.linalg.norm(E, axis=1, keepdims=True)).astype(np.float16)
    Ev = E16.astype(np.float32)
    Qb = np.stack([oracle.synth_matrix(1700 + i, nq, dim) for i in range(n_queries)])
    split = int(off[n_chunks // 2])
    idx = raglite_amd.DeviceIndex(E16[:split], off[: n_chunks // 2 + 1], metric="dot", storage="f16")
    idx.append(E16[split:], np.diff(off[n_chunks // 2:]))
    k = 40
    bs, bc = idx.maxsim_topk_batch(Qb, k)
    for i in range(n_queries):
        ref = oracle.maxsim_scores(Ev, off, Qb[i], np.float64)
        assert_topk_close(bs[i], bc[i], ref, k, 1e-5 * float(np.abs(ref).max()))
    dead = np.unique(bc[:, 0])
    idx.delete_chunks(dead)
    bs2, bc2 = idx.maxsim_topk_batch(Qb, k)
    assert not np.isin(bc2, dead).any()
    for i in (0, n_queries - 1):
        ref = oracle.maxsim_scores(Ev, off, Qb[i], np.float64).copy()
        ref[dead] = -np.inf
        assert_topk_close(bs2[i], bc2[i], ref, k, 1e-5 * float(np.abs(ref[np.isfinite(ref)]).max()))
    idx.close()
