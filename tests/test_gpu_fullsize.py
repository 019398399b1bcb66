"""BASELINE.json's full sizes on the GPU (configs 2, 3, 4, 5 and the metric shape), checked through size-independent
properties, against float64 references computed by an independent implementation (PyTorch-ROCm's fp64 GEMM /
reductions on the GPU: the CPU oracle would take minutes on every chunk of 1 M x 1024) and against the NumPy oracle on
sampled queries / spans.  Also: concurrent callers (the reference calls the path from worker threads).
"""

import threading

import numpy as np
import pytest

import raglite_amd
from bench import chunk_offsets

pytestmark = pytest.mark.gpu
N, D = 1_000_000, 1024


@pytest.fixture(scope="module")
def corpus():
    import torch

    raglite_amd.set_device(0)
    E = torch.empty((N, D), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=6)
    yield torch, E
    del E
    torch.cuda.empty_cache()


def _check_topk_against(torch, got_s, got_i, ref_scores, k, atol):
    """Tie-aware: returned scores match the reference at the returned ids (abs tol), are sorted, ids distinct, and
    nothing better by more than 2*atol was missed."""
    got_s, got_i = got_s.double(), got_i.long()
    assert got_i.unique().numel() == k
    ref = ref_scores.double()
    assert float((got_s - ref[got_i]).abs().max()) <= atol
    assert bool((got_s[1:] <= got_s[:-1]).all())
    better = (ref > got_s[-1] + 2 * atol).nonzero().flatten()
    assert set(better.tolist()) <= set(got_i.tolist())


def _segment_max(torch, S, lengths):
    try:
        return torch.segment_reduce(S, "max", lengths=lengths, axis=0)
    except Exception:  # noqa: BLE001 - older builds: scatter form
        ids = torch.repeat_interleave(torch.arange(len(lengths), device=S.device), lengths)
        out = torch.full((len(lengths), S.shape[1]), float("-inf"), device=S.device, dtype=S.dtype)
        return out.scatter_reduce(0, ids[:, None].expand(-1, S.shape[1]), S, "amax")


def test_fullsize_cosine_top100_and_shard_merge(corpus):
    """cfg 2: 1 M x 1024 fp32, single-query cosine top-100; merge of two half-corpus searches == full search."""
    torch, E = corpus
    q = torch.empty((3, D), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q, seed=61)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    norms = E.norm(dim=1)
    for b in range(2):
        s, r = idx.search_rows(q[b], 100)
        cos = (E @ q[b]) / (norms * q[b].norm())
        _check_topk_against(torch, s, r, cos, 100, 1e-4)  # the 1e-4 fp32 bar of north_star
        s2, r2 = idx.search_rows(q[b], 100)
        assert torch.equal(s, s2) and torch.equal(r, r2)  # deterministic
    full_s, full_r = idx.search_rows(q, 100)  # batch of 3 (VALU path, 2 + 1 passes)
    half = N // 2 + 7
    lo = raglite_amd.DeviceIndex(E[:half], metric="cosine")
    hi = raglite_amd.DeviceIndex(E[half:], metric="cosine")
    s0, r0 = lo.search_rows(q, 100)
    s1, r1 = hi.search_rows(q, 100)
    ms, mr = raglite_amd.merge_topk(torch.stack([s0, s1]), torch.stack([r0, r1 + half]).int(), 100)
    assert torch.equal(ms, full_s) and torch.equal(mr, full_r)
    for i in (idx, lo, hi):
        i.close()


def test_fullsize_batched_cosine_mfma_path(corpus):
    """33 queries at once take the MFMA tile kernel (dim 1024, B > 4) + metric transform."""
    torch, E = corpus
    Q = torch.empty((33, D), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=62)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    S, R = idx.search_rows(Q, 100)
    norms = E.norm(dim=1)
    for b in (0, 16, 32):
        cos = (E @ Q[b]) / (norms * Q[b].norm())
        _check_topk_against(torch, S[b], R[b], cos, 100, 1e-4)
    idx.close()


def test_fullsize_gemm_path_cfg5_shape(corpus):
    """cfg 5's per-GPU shape class: hundreds of queries against the full 1 M x 1024 corpus through score_gemm.hip."""
    torch, E = corpus
    B = 300
    Q = torch.empty((B, D), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=64)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    S, R = idx.search_rows(Q, 100)
    norms = E.norm(dim=1)
    for b in (0, 127, 128, 299):
        cos = (E @ Q[b]) / (norms * Q[b].norm())
        _check_topk_against(torch, S[b], R[b], cos, 100, 1e-4)
    # the merge of two half-corpus searches equals the full search, bit for bit (scores are position-independent)
    half = N // 2 + 77
    lo = raglite_amd.DeviceIndex(E[:half], metric="cosine")
    hi = raglite_amd.DeviceIndex(E[half:], metric="cosine")
    s0, r0 = lo.search_rows(Q, 100)
    s1, r1 = hi.search_rows(Q, 100)
    ms, mr = raglite_amd.merge_topk(torch.stack([s0, s1]), torch.stack([r0, r1 + half]).int(), 100)
    assert torch.equal(ms, S) and torch.equal(mr, R)
    for i in (idx, lo, hi):
        i.close()


def test_fullsize_maxsim_32x1m(corpus):
    """The metric shape: 32 query vectors x 1 M chunk vectors, ragged chunks, exact top-100 -- single-query kernel, the
    two-queries-per-pass kernel and the eight-queries-per-pass kernel over the pre-split corpus image, all against a
    float64 reference of EVERY chunk score (torch on the GPU, fp64 GEMM) at 2e-6 of the score scale (the kernels deliver
    ~3e-7; a dropped `lo` term of the fp16 split would show up at 5e-4)."""
    torch, E = corpus
    off = chunk_offsets(N)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    assert idx.arithmetic == "f16_split"
    Qb = torch.empty((9, 32, D), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Qb, seed=63)
    Q = Qb[0]
    lengths = torch.as_tensor(np.diff(off), device="cuda")
    E64 = E.double()
    refs = [_segment_max(torch, E64 @ Qb[i].double().T, lengths).sum(dim=1) for i in range(9)]  # (n_chunks,) float64 each
    del E64
    ref = refs[0]
    scale = ref.abs().max().item()  # |score| ~ 5e2 here (un-normalised U(-1,1) rows)
    tol = 2e-6 * scale
    scores = idx.maxsim_scores(Q)
    assert float((scores.double() - ref).abs().max()) <= tol
    s, c = idx.maxsim_topk(Q, 100)  # one query: ranked from the HI plane, re-scored exactly over the rows (round 6)
    assert idx.filter_stats()["kind"] == "maxsim_batch_hi" and not idx.filter_stats()["fallback"]
    _check_topk_against(torch, s, c, ref, 100, tol)
    with idx.options(hi_few=0):  # the streaming kernel over the fp32 rows it replaces: the same chunks
        s_rows, c_rows = idx.maxsim_topk(Q, 100)
    assert set(c_rows.tolist()) == set(c.tolist())
    # top-k of OUR scores is exact (bitwise) with ties to the lowest chunk ordinal
    order = torch.sort(scores, descending=True, stable=True).indices[:100]
    assert torch.equal(c_rows.long(), order) and torch.equal(s_rows, scores[order])
    # size-independent properties
    assert torch.equal(idx.maxsim_scores(2.0 * Q), 2.0 * scores)  # exact linearity in powers of two
    assert torch.equal(idx.maxsim_scores(Q), scores)  # deterministic
    sb, cb = idx.maxsim_topk_batch(torch.stack([Q, Q]), 100)  # two queries: the pair kernel, same bits as the single-query one
    assert torch.equal(sb[0], s) and torch.equal(sb[1], s) and torch.equal(cb[1], c)
    # nine queries: one pass of eight (maxsim_gemm_kernel) + one single; every query against its float64 reference
    s9, c9 = idx.maxsim_topk_batch(Qb, 100)
    for i in range(9):
        _check_topk_against(torch, s9[i], c9[i], refs[i], 100, tol)
    assert torch.equal(idx.maxsim_topk_batch(Qb, 100)[0], s9)  # deterministic
    s9x, c9x = idx.maxsim_topk_batch(2.0 * Qb, 100)
    assert torch.equal(s9x, 2.0 * s9) and torch.equal(c9x, c9)  # exact linearity in powers of two
    # shard by chunk: merged local top-k == global top-k (bitwise), single query and the eight-query pass
    cut = len(off) // 3
    a = raglite_amd.DeviceIndex(E[: off[cut]], off[: cut + 1], metric="dot")
    b = raglite_amd.DeviceIndex(E[off[cut]:], off[cut:] - off[cut], metric="dot")
    sa, ca = a.maxsim_topk(Q, 100)
    sb2, cb2 = b.maxsim_topk(Q, 100)
    ms, mc = raglite_amd.merge_topk(torch.stack([sa, sb2])[:, None, :], torch.stack([ca, cb2 + cut]).int()[:, None, :], 100)
    assert torch.equal(ms[0], s) and torch.equal(mc[0], c)
    sa8, ca8 = a.maxsim_topk_batch(Qb[:8], 100)
    sb8, cb8 = b.maxsim_topk_batch(Qb[:8], 100)
    ms8, mc8 = raglite_amd.merge_topk(torch.stack([sa8, sb8]), torch.stack([ca8, cb8 + cut]).int(), 100)
    assert torch.equal(ms8, s9[:8]) and torch.equal(mc8, c9[:8])
    # the exact fp32 chain on the same queries: same chunks wherever the float64 scores are further apart than the bar
    idx.set_exact_fp32(True)
    se, ce = idx.maxsim_topk_batch(Qb, 100)
    for i in range(9):
        _check_topk_against(torch, se[i], ce[i], refs[i], 100, tol)
    for i in (idx, a, b):
        i.close()


def test_fullsize_cfg3_rerank_4096_queries():
    """BASELINE cfg 3 at full scale: 4096 independent queries x 32 vectors, each against its own 256 candidate chunks of
    64 vectors, d = 128, unit rows (ColBERT convention), fp32- and fp16-stored corpus.  Sampled queries against the
    float64 oracle (north star: within 1e-4; unit rows give |score| <= 32), every query against single-query calls
    (bitwise) and under a permutation of its candidates (bitwise)."""
    import torch

    from oracle import oracle

    raglite_amd.set_device(0)
    d, nq, n_cand, rows, n_chunks, nb = 128, 32, 256, 64, 16384, 4096
    E = torch.empty((n_chunks * rows, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=3)
    E /= E.norm(dim=1, keepdim=True)
    off = np.arange(0, n_chunks * rows + 1, rows, dtype=np.int64)
    Q = torch.empty((nb, nq, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=30)
    Q /= Q.norm(dim=2, keepdim=True)
    g = torch.Generator(device="cuda").manual_seed(33)
    cand = torch.randint(0, n_chunks, (nb, n_cand), device="cuda", dtype=torch.int32, generator=g)
    for storage in ("f32", "f16"):
        Es = E.half() if storage == "f16" else E
        idx = raglite_amd.DeviceIndex(Es, off, metric="dot", storage=storage)
        got = idx.maxsim_rerank(Q, cand)
        assert got.shape == (nb, n_cand) and bool(torch.isfinite(got).all())
        Eh = Es.float().cpu().numpy()
        for b in (0, 1, 1023, 2048, 4095):
            want = oracle.maxsim_candidates(Eh, off, Q[b].cpu().numpy(), cand[b].cpu().numpy(), np.float64)
            np.testing.assert_allclose(got[b].cpu().numpy(), want, rtol=0, atol=1e-4 / 4)
        for b in (7, 4000):  # a query scores the same alone as in the batch of 4096
            assert torch.equal(idx.maxsim_rerank(Q[b : b + 1], cand[b : b + 1])[0], got[b])
        perm = torch.randperm(n_cand, device="cuda", generator=g)
        assert torch.equal(idx.maxsim_rerank(Q, cand[:, perm]), got[:, perm])
        idx.close()


def test_fullsize_cfg4_pool_index_adapter_search():
    """BASELINE cfg 4 at full scale, end to end on the device: 100 000 sentences of 4..60 token rows (3.2 M x 1024 fp32)
    -> late-chunking pool + L2-normalise + fp16 (src/raglite/_embed.py:131-140) -> 20 000 multi-vector chunks
    (src/raglite/_split_chunks.py:116-122) -> query adapter at B = 1 and B = 1000 (src/raglite/_search.py:58-62) ->
    two-stage chunk search (:66-79,143-149).  Sampled spans / queries against the oracle."""
    import torch

    from oracle import oracle

    raglite_amd.set_device(0)
    d, S = 1024, 100_000
    rng = np.random.default_rng(4)
    lens = rng.integers(4, 61, size=S)
    ends = np.cumsum(lens)
    begins = ends - lens
    T = int(ends[-1])
    tokens = torch.empty((T, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(tokens, seed=4)
    _, emb16 = raglite_amd.pool_norm(tokens, torch.as_tensor(begins, device="cuda"), torch.as_tensor(ends, device="cuda"))
    assert emb16.shape == (S, d) and emb16.dtype == torch.float16
    norms = emb16.float().norm(dim=1)
    assert float((norms - 1).abs().max()) < 1e-3  # unit-norm rtol 1e-3: the reference's own test (tests/test_embed.py:24-26)
    e16 = emb16.cpu().numpy()
    for sidx in rng.choice(S, size=500, replace=False):
        rows_h = tokens[int(begins[sidx]) : int(ends[sidx])].cpu().numpy()
        _, ref16 = oracle.pool_norm_cast(rows_h, np.array([0]), np.array([len(rows_h)]))
        diff = np.abs(ref16[0].view(np.int16).astype(np.int32) - e16[sidx].view(np.int16).astype(np.int32))
        assert diff.max() <= 1, (sidx, diff.max())  # <= 1 fp16 ulp (fp32 inputs pooled in fp64 on both sides; the divide differs)
    del tokens
    torch.cuda.empty_cache()
    # 100 k rows -> 20 k chunks of 1..9 rows
    off = np.concatenate(([0], np.cumsum(rng.integers(1, 10, size=S // 4))))
    off = np.concatenate((off[off < S], [S])).astype(np.int64)
    assert 18_000 < len(off) - 1 < 22_000
    E = emb16.float()  # the fp32 widening of the stored fp16 values, as the BASELINE configs define the corpus
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    A = np.linalg.qr(rng.standard_normal((d, d)))[0].astype(np.float32)  # orthogonal adapter (the cosine solution is U V^T)
    Qraw = rng.standard_normal((1000, d)).astype(np.float32)
    Qa = raglite_amd.adapter_apply(torch.as_tensor(A, device="cuda"), torch.as_tensor(Qraw, device="cuda"))
    want = Qraw.astype(np.float64) @ A.astype(np.float64).T
    np.testing.assert_allclose(Qa.cpu().numpy(), want, rtol=0, atol=2e-5)  # |q'| ~ 32, fp32 accumulation
    q1 = raglite_amd.adapter_apply(A, Qraw[0])
    np.testing.assert_allclose(q1, want[0], rtol=0, atol=2e-5)
    num_hits, k = oracle.num_hits(10), 10
    s, c, cnt = idx.search_chunks(Qa, num_hits, k)
    s, c, cnt = s.cpu().numpy(), c.cpu().numpy(), cnt.cpu().numpy()
    Eh = E.cpu().numpy()
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    for b in (0, 333, 999):
        ws, wc = oracle.search_chunks(Eh, r2c, Qa[b].cpu().numpy(), num_hits, k, "cosine", np.float64)
        assert int(cnt[b]) == len(wc)
        np.testing.assert_allclose(s[b, : len(wc)], ws, rtol=0, atol=1e-4 / 4)
        assert set(c[b, : len(wc)].tolist()) == set(wc.tolist()) or np.abs(np.diff(ws)).min() < 1e-6
    s1, c1, cnt1 = idx.search_chunks(Qa[0], num_hits, k)  # B = 1: the scan path; same chunks as the batched GEMM path
    assert c1.cpu().numpy()[: int(cnt1)].tolist() == c[0, : int(cnt[0])].tolist()
    idx.close()


def test_fullsize_cfg5_shard_1000_queries():
    """BASELINE cfg 5, one of the 8 shards at full scale: 1000 queries x 1.25 M x 1024 fp32, cosine exact top-100, through
    score_gemm256_kernel.  Sampled queries against float64 (torch fp64 on the GPU) and against the fp32 NumPy oracle on
    the host; two half-shards merged == the shard, bit for bit (what the 8-rank all-gather merge relies on)."""
    import torch

    from oracle import oracle

    raglite_amd.set_device(0)
    n, d, B, k = 1_250_000, 1024, 1000, 100
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=5)
    Q = torch.empty((B, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=50)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    S, R = idx.search_rows(Q, k)
    assert S.shape == (B, k) and bool((S[:, 1:] <= S[:, :-1]).all())
    n64 = E.double().norm(dim=1)
    for b in (0, 255, 256, 511, 999):  # both sides of the 256-query tile boundaries
        cos = (E.double() @ Q[b].double()) / (n64 * Q[b].double().norm())
        _check_topk_against(torch, S[b], R[b], cos, k, 2e-6)  # cosine of U(-1,1) rows: |score| <= 0.2, error ~1e-7
    del n64
    Eh = E.cpu().numpy()
    for b in (3, 777):
        ws, wr = oracle.search_rows(Eh, Q[b].cpu().numpy(), k, "cosine", np.float32)
        assert len(set(wr.tolist()) & set(R[b].cpu().numpy().tolist())) >= k - 1  # an fp32 tie at the 100th place may swap one
        np.testing.assert_allclose(S[b].cpu().numpy(), ws, rtol=0, atol=2e-6)
    del Eh
    half = n // 2 + 77
    lo = raglite_amd.DeviceIndex(E[:half], metric="cosine")
    hi = raglite_amd.DeviceIndex(E[half:], metric="cosine")
    s0, r0 = lo.search_rows(Q, k)
    s1, r1 = hi.search_rows(Q, k)
    ms, mr = raglite_amd.merge_topk(torch.stack([s0, s1]), torch.stack([r0, r1 + half]).int(), k)
    assert torch.equal(ms, S) and torch.equal(mr, R)
    for i in (idx, lo, hi):
        i.close()


def test_concurrent_callers(corpus):
    """Four threads share one index and the pooling entry point (src/raglite/_insert.py:208-237 uses <= 4 workers)."""
    torch, E = corpus
    sub = E[:200_000]
    idx = raglite_amd.DeviceIndex(sub, metric="cosine")
    rng = np.random.default_rng(0)
    queries = rng.standard_normal((8, D)).astype(np.float32)
    tokens = rng.standard_normal((4000, D)).astype(np.float32)
    ends = np.arange(40, 4001, 40)
    begins = ends - 40
    serial_search = [idx.search_rows(q, 50) for q in queries]
    serial_pool = raglite_amd.pool_norm(tokens, begins, ends)[1]
    errors, results = [], {}

    def worker(tid):
        try:
            raglite_amd.set_device(0)
            out = []
            for rep in range(3):
                for j, q in enumerate(queries):
                    out.append((j, idx.search_rows(q, 50)))
                p = raglite_amd.pool_norm(tokens, begins, ends)[1]
                assert np.array_equal(p.view(np.uint16), serial_pool.view(np.uint16))
            results[tid] = out
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for out in results.values():
        for j, (s, r) in out:
            assert np.array_equal(s, serial_search[j][0]) and np.array_equal(r, serial_search[j][1])
    idx.close()
