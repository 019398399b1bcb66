"""GPU parity of the sixteen-queries-per-pass approximate MaxSim kernel (raglite_amd/csrc/maxsim_pp.hip), the first stage of
`rl_maxsim_topk_batch`'s bound-filtered pipeline, through `rl_maxsim_approx_scores`:

score[c] ~ sum_i max_{j in chunk c} Q[i].D[j] from the hi halves of corpus and queries -- the multi-vector generalisation of
`/root/reference/src/raglite/_search.py:143-149` behind the reranker plugin call (:394-396).

Bars: EVERY (query, chunk) score against the eight-queries-per-pass kernel of maxsim_gemm.hip -- the same products and the same sums
over K; the 32 per-vector maxima of a chunk are added in another order (a tree over lanes), so: bit-identical on integer-valued data in
every chunk layout, within 4 float32 ulps of the score scale on float data; integer data bit-identical to the oracle for every chunk
(hi halves of small integers are the integers); on float data |approximate - float64| <= the bound the pipeline builds its candidate
window on, for every chunk; and the whole pipeline over either kernel returns the same bits (its scores come from the exact
re-scoring)."""

import os

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests.util import ragged_offsets

pytestmark = pytest.mark.gpu



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


@pytest.mark.parametrize("kind", ["small_int", "uniform"])
@pytest.mark.parametrize("n,dim,nq,n_queries,layout", [
    (70_003, 1024, 32, 16, "ragged"),      # the headline shape class; n not a multiple of 16
    (70_000, 1024, 17, 21, "ragged"),      # a second, partial pass (5 queries: waves without a query, one with a single query)
    (70_000, 1024, 1, 3, "ragged"),        # one query vector; three queries
    (70_000, 1024, 16, 9, "ones"),         # every row its own chunk: 128 stores per tile and wave
    (66_000, 1024, 32, 16, "giant"),       # chunks of 1 000 rows: a chunk spans eight tiles and workgroup boundaries
    (140_000, 512, 32, 19, "ragged"),      # dim 512: 16 slabs per tile
    (270_000, 256, 9, 16, "ragged"),       # dim 256: 8 slabs per tile (the smallest the kernel takes)
])
def test_pp_against_the_eight_query_kernel(n, dim, nq, n_queries, layout, kind):
    torch = _torch()
    rng = np.random.default_rng(n + nq)
    off = {"ragged": lambda: ragged_offsets(rng, n, 1, 15), "ones": lambda: None,
           "giant": lambda: np.concatenate((np.arange(0, n, 1000), [n])).astype(np.int64)}[layout]()
    E = _corpus(torch, n, dim, seed=900 + nq, kind=kind)
    Q = _queries(torch, n_queries, nq, dim, seed=901 + nq, kind=kind)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    a, ma = idx.maxsim_approx_scores(Q, kernel=0)
    b, mb = idx.maxsim_approx_scores(Q, kernel=1)
    if kind == "small_int":  # every partial sum is an integer below 2^24: the order of the additions cannot matter
        assert torch.equal(a, b), f"{int((a != b).sum())} of {a.numel()} scores differ"
    else:
        ulp = float(b.abs().max()) * 2.0 ** -23
        assert float((a - b).abs().max()) <= 4 * ulp, (float((a - b).abs().max()), ulp)
    assert torch.equal(ma, mb) and bool((ma > 0).all())
    a2, _ = idx.maxsim_approx_scores(Q, kernel=0)
    assert torch.equal(a, a2)  # deterministic
    idx.close()


def test_pp_integer_data_is_the_oracle_bit_for_bit():
    torch = _torch()
    n, dim, nq, n_queries = 70_000, 1024, 32, 5
    rng = np.random.default_rng(7)
    off = ragged_offsets(rng, n, 1, 15)
    E = oracle.synth_matrix(20_000, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(20_100 + i, nq, dim, "small_int") for i in range(n_queries)])
    idx = raglite_amd.DeviceIndex(torch.from_numpy(E).cuda(), off, metric="dot")
    got, _ = idx.maxsim_approx_scores(torch.from_numpy(Qb).cuda(), kernel=0)
    got = got.cpu().numpy()
    for i in (0, 2, 4):
        want = oracle.maxsim_scores(E, off, Qb[i], np.float32)
        assert np.array_equal(got[i], want), i
    idx.close()


@pytest.mark.parametrize("shape", ["uniform", "unit_fp16"])
def test_pp_error_stays_inside_the_bound_for_every_chunk(shape):
    """The bound the candidate window is built on, measured: |approximate - float64| <= m for EVERY chunk of every query."""
    torch = _torch()
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import bench_configs

    n, dim, nq, n_queries = 70_000, 1024, 32, 16
    rng = np.random.default_rng(11)
    off = ragged_offsets(rng, n, 1, 15)
    E = _corpus(torch, n, dim, seed=31)
    Q = _queries(torch, n_queries, nq, dim, seed=32)
    if shape == "unit_fp16":
        E = torch.nn.functional.normalize(E, dim=1).half().float()
        Q = torch.nn.functional.normalize(Q, dim=2).half().float()
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    got, m = idx.maxsim_approx_scores(Q, kernel=0)
    ref = bench_configs.maxsim_scores_f64(E, off, Q)
    err = (got.double() - ref).abs().max(dim=1).values
    assert bool((err <= m.double()).all()), (err.tolist(), m.tolist())
    assert float((err / m.double()).max()) < 0.5  # Cauchy-Schwarz is far from tight on such data: the bound has room
    idx.close()


def test_pipeline_over_either_kernel_returns_the_same_bits():
    torch = _torch()
    n, dim, nq, n_queries, k = 70_000, 1024, 32, 37, 100
    rng = np.random.default_rng(3)
    off = ragged_offsets(rng, n, 1, 15)
    E = _corpus(torch, n, dim, seed=41)
    Q = _queries(torch, n_queries, nq, dim, seed=42)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s0, c0 = idx.maxsim_topk_batch(Q, k)
    assert idx.filter_stats()["kind"] == "maxsim_batch_hi" and not idx.filter_stats()["fallback"]
    with idx.options(pp_pass=0):
        s1, c1 = idx.maxsim_topk_batch(Q, k)
    assert torch.equal(s0, s1) and torch.equal(c0, c1)
    with idx.options(hi_maxsim=0):  # the full-precision passes: same chunks, scores to the last bits of the split arithmetic
        s2, c2 = idx.maxsim_topk_batch(Q, k)
    assert torch.equal(c0, c2)
    assert float((s0 - s2).abs().max()) <= 2e-6 * float(s0.abs().max())
    idx.close()


def test_pp_after_append_and_on_a_sub_range():
    """The image grows with the index (rl_index_append) and the kernel's row ranges follow the chunk structure."""
    torch = _torch()
    n, dim, nq, n_queries = 66_000, 1024, 32, 16
    rng = np.random.default_rng(5)
    E = _corpus(torch, n + 5_000, dim, seed=51)
    off = ragged_offsets(rng, n, 1, 15)
    extra = ragged_offsets(rng, 5_000, 1, 15)
    Q = _queries(torch, n_queries, nq, dim, seed=52)
    idx = raglite_amd.DeviceIndex(E[:n].clone(), off, metric="dot")
    idx.append(E[n:], np.diff(extra))
    whole = raglite_amd.DeviceIndex(E, np.concatenate((off, off[-1] + extra[1:])), metric="dot")
    a, _ = idx.maxsim_approx_scores(Q, kernel=0)
    b, _ = whole.maxsim_approx_scores(Q, kernel=0)
    assert torch.equal(a, b)
    idx.close()
    whole.close()


# ---- adversarial inputs for the bound (round 4) -----------------------------------------------------------------------------------------
# The candidate window of the headline pipeline is exact only if |approximate - exact| <= m holds for EVERY chunk on ANY data.  The random
# corpora above sit far inside the bound (Cauchy-Schwarz is loose on them); these are built to come as close to it as the construction allows:
#   aligned     every row is e = H + 0.45 q with H fp16-representable at the index' scale (scale = 1: one element is 2^13) and the SAME q as
#               all 32 vectors of every query: what the hi halves drop, e_lo = 0.45 q, is parallel to every query vector -- |q . e_lo| =
#               |q| |e_lo| for every pair, every chunk, with one sign: the per-pair bound is met with equality and the 32 errors add up
#   positive    corpus and queries all positive (no cancellation in any dot product)
#   subnormal   half the rows 2^-9 of the others with heavy-tailed elements: their hi halves are fp16 subnormals or zero
#   giant       one row 500 x the norm of the rest: the bound takes the MAXIMUM of |e_lo| and |e| over the rows, so m is 500 x what the
#               other rows need -- valid but useless: the window holds every chunk, the lists overflow, and the guarded full-precision
#               passes must answer (flag asserted)
# Asserted for each: err <= m per chunk (err / m is printed: how close the worst case comes), and the pipeline's top-k equals the
# full-precision passes' (same chunks; scores to the last bits of the two arithmetics) and the float64 ranking.


def _adversarial(torch, kind, n, dim, n_queries, nq):
    g = torch.Generator(device="cuda").manual_seed(1234)
    U = lambda *shape: torch.rand(*shape, generator=g, device="cuda") * 2 - 1  # noqa: E731
    if kind == "aligned":
        levels = torch.tensor([-1.0, -0.75, -0.5, -0.25, 0.25, 0.5, 0.75, 1.0], device="cuda")
        q = levels[torch.randint(0, 8, (n_queries, 1, dim), generator=g, device="cuda")]  # fp16-representable at any power-of-two scale
        Q = q.expand(n_queries, nq, dim).contiguous()
        H = torch.randint(1024, 2048, (n, dim), generator=g, device="cuda").float() * torch.where(U(n, dim) < 0, -1.0, 1.0)
        E = H + 0.45 * Q[0, 0][None, :]  # aligned with query 0's vectors (the other queries see a generic residual)
        E[0, 0] = 8192.0  # pins the index' scale to 1: fp16 spacing is 1 in [1024, 2048), so hi = H exactly
        return E.contiguous(), Q
    if kind == "positive":
        return U(n, dim).abs().contiguous(), U(n_queries, nq, dim).abs().contiguous()
    if kind == "subnormal":
        E = U(n, dim)
        small = torch.arange(n, device="cuda") % 2 == 1
        E[small] = (E[small].sign() * E[small].abs() ** 6) * 2.0 ** -9
        return E.contiguous(), U(n_queries, nq, dim).contiguous()
    if kind == "giant":
        E = U(n, dim)
        E[n // 2] *= 500.0
        return E.contiguous(), U(n_queries, nq, dim).contiguous()
    raise AssertionError(kind)


@pytest.mark.parametrize("dim", [1024, 3072])  # (3072, round 6: the bound's rounding term grows with dim -- api.hip: sum_eps)
@pytest.mark.parametrize("kind", ["aligned", "positive", "subnormal"])
def test_pp_bound_holds_on_adversarial_data(kind, dim):
    torch = _torch()
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import bench_configs

    n, nq, n_queries, k = (70_000 if dim == 1024 else 24_000), 32, 16, 50
    rng = np.random.default_rng(17)
    off = ragged_offsets(rng, n, 1, 15)
    E, Q = _adversarial(torch, kind, n, dim, n_queries, nq)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    assert idx.arithmetic == "f16_split"
    got, m = idx.maxsim_approx_scores(Q, kernel=0)
    got8, m8 = idx.maxsim_approx_scores(Q, kernel=1)
    ref = bench_configs.maxsim_scores_f64(E, off, Q)
    for name, a in (("sixteen-query kernel", got), ("eight-query kernel", got8)):
        err = (a.double() - ref).abs().max(dim=1).values
        ratio = (err / m.double())
        print(f"[{kind}] {name}: worst err / m = {float(ratio.max()):.3f} (query {int(ratio.argmax())}), m = {float(m[int(ratio.argmax())]):.4g}, "
              f"score scale {float(ref.abs().max()):.4g}")
        assert bool((err <= m.double()).all()), (kind, name, err.tolist(), m.tolist())
    if kind == "aligned":  # query 0 meets the per-pair bound with equality: the measured error must be most of what the e_lo term allows
        e_lo_term = 0.45 * float(Q[0, 0].norm()) * float(Q[0].norm(dim=1).sum())
        err0 = float((got[0].double() - ref[0]).abs().max())
        assert err0 >= 0.98 * e_lo_term, (err0, e_lo_term)
    s, c = idx.maxsim_topk_batch(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi"
    with idx.options(hi_maxsim=0):
        s0, c0 = idx.maxsim_topk_batch(Q, k)
    ref_top = torch.topk(ref, k, dim=1)
    scale = float(ref.abs().max())
    for b in range(n_queries):
        want = set(ref_top.indices[b].tolist())
        # the float64 ranking, up to swaps among scores closer than the fp32 arithmetic resolves
        for got_c, got_s in ((c[b], s[b]), (c0[b], s0[b])):
            extra = set(got_c.tolist()) ^ want
            if extra:
                kth = float(ref_top.values[b, -1])
                assert all(abs(float(ref[b, e]) - kth) <= 4e-6 * scale for e in extra), (kind, b, sorted(extra)[:6])
            assert float((got_s.double() - ref[b, got_c.long()]).abs().max()) <= 2e-6 * scale
    idx.close()


def test_giant_row_defeats_the_bound_and_the_fallback_answers():
    torch = _torch()
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))
    import bench_configs

    n, dim, nq, n_queries, k = 70_000, 1024, 32, 8, 20
    rng = np.random.default_rng(19)
    off = ragged_offsets(rng, n, 1, 15)
    E, Q = _adversarial(torch, "giant", n, dim, n_queries, nq)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    idx.set_option("lazy_images", 0)  # (the pre-split image from the start: the guarded fallback is the eight-query full-precision pass)
    assert idx.arithmetic == "f16_split"  # (500 x is inside the 2^10 window that keeps the split arithmetic)
    got, m = idx.maxsim_approx_scores(Q, kernel=0)
    ref = bench_configs.maxsim_scores_f64(E, off, Q)
    err = (got.double() - ref).abs().max(dim=1).values
    print(f"[giant] worst err / m = {float((err / m.double()).max()):.3f}; m / score spread = {float(m.max() / ref.std(dim=1).min()):.1f}")
    assert bool((err <= m.double()).all())
    s, c = idx.maxsim_topk_batch(Q, k)
    st = idx.filter_stats()
    assert st["kind"] == "maxsim_batch_hi" and st["fallback"], st  # the window holds (nearly) every chunk: the lists overflow
    with idx.options(hi_maxsim=0):
        s0, c0 = idx.maxsim_topk_batch(Q, k)
    assert torch.equal(c, c0) and torch.equal(s, s0)  # the SAME passes answered both times
    idx.close()


def test_approx_scores_refuse_an_index_with_an_empty_chunk():
    """Both pass kernels find a chunk by counting chunk ends; an empty chunk would shift every score behind it.  The header promises
    RL_ERR_UNSUPPORTED (ValueError) -- and the batch search over the same index takes the streaming kernels and stays correct."""
    torch = _torch()
    n, dim, nq = 70_000, 1024, 8
    rng = np.random.default_rng(23)
    off = ragged_offsets(rng, n, 1, 15, empty_every=1000)
    assert (np.diff(off) == 0).any()
    E = _corpus(torch, n, dim, seed=61, kind="small_int")
    Q = _queries(torch, 4, nq, dim, seed=62, kind="small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    for kernel in (0, 1):
        with pytest.raises(ValueError, match="empty chunk"):
            idx.maxsim_approx_scores(Q, kernel=kernel)
    s, c = idx.maxsim_topk_batch(Q, 10)
    Eh, Qh = E.cpu().numpy(), Q.cpu().numpy()
    for b in range(4):
        ws, wc = oracle.maxsim_topk(Eh, off, Qh[b], 10, np.float32)
        assert np.array_equal(c[b].cpu().numpy(), wc) and np.array_equal(s[b].cpu().numpy(), ws)
    idx.close()


def test_all_passes_in_one_launch_equal_one_launch_per_pass():
    """Round 4: a batch's passes are grid rows of ONE launch.  40 queries (two and a half passes) against the same queries in groups of at
    most sixteen: the same bits, chunk for chunk."""
    torch = _torch()
    n, dim, nq = 70_000, 1024, 32
    rng = np.random.default_rng(77)
    off = ragged_offsets(rng, n, 1, 15)
    idx = raglite_amd.DeviceIndex(_corpus(torch, n, dim, seed=881), off, metric="dot")
    Q = _queries(torch, 40, nq, dim, seed=882)
    whole, bound = idx.maxsim_approx_scores(Q, kernel=0)
    for lo, hi in ((0, 16), (16, 32), (32, 40), (3, 12)):
        part, b = idx.maxsim_approx_scores(Q[lo:hi].contiguous(), kernel=0)
        assert torch.equal(part, whole[lo:hi]) and torch.equal(b, bound[lo:hi])
    idx.close()


def test_results_staged_in_lds_are_flushed_when_the_buffer_fills():
    """The pass keeps a workgroup's chunk scores in LDS (1 536 per query of a wave) and writes them out when the workgroup is done -- or when
    the buffer is nearly full: 420 000 one-row chunks put ~1 640 chunks into every workgroup's range, so each flushes in the middle of its
    rows.  Integer data: the oracle bit for bit (a score written twice, or into the neighbour's ordinals, would show)."""
    n, dim, nq = 420_000, 256, 5
    off = np.arange(n + 1, dtype=np.int64)
    E = oracle.synth_matrix(889, n, dim, "small_int")
    Qb = np.stack([oracle.synth_matrix(890 + i, nq, dim, "small_int") for i in range(19)])  # a full pass and a partial one (3 of 16 queries)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    approx, _ = idx.maxsim_approx_scores(Qb, kernel=0)
    for b in (0, 15, 16, 18):
        assert np.array_equal(approx[b].astype(np.float64), oracle.maxsim_scores(E, off, Qb[b], np.float64)), b
    s, c = idx.maxsim_topk_batch(Qb, 50)
    for b in (3, 17):
        ws, wc = oracle.maxsim_topk(E, off, Qb[b], 50, np.float32)
        assert np.array_equal(c[b], wc) and np.array_equal(s[b], ws)
    idx.close()
