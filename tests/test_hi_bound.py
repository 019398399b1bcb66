"""CPU property tests of the error bound behind the half-bytes searches (`api.hip: search_rows_hi`,
`rl_maxsim_topk_batch`): ranking on the HI halves of the fp16 split is only a pruning step if the bound on what the LO halves
contribute is rigorous.  Here the split, the bound and the candidate-set argument are restated in NumPy float64 and checked on
random and on adversarial inputs -- the device kernels are checked against the full-precision path in tests/test_gpu_hi_*.py.

Reference semantics being preserved: `ORDER BY dist LIMIT k` (`/root/reference/src/raglite/_search.py:69-79`) and the
per-chunk maximum (`:143-149`), ranked exactly."""

import numpy as np
import pytest


def _split_hi(E: np.ndarray) -> tuple[np.ndarray, np.ndarray, float]:
    """hi = fp16(e * scale) rounded TOWARD ZERO, lo = e - hi / scale, scale = the index' power of two (max |e| -> [2^13, 2^14))."""
    mx = float(np.abs(E).max())
    scale = 2.0 ** (14 - int(np.floor(np.log2(mx)) + 1)) if mx > 0 else 1.0
    x = E.astype(np.float64) * scale
    h = x.astype(np.float16).astype(np.float64)  # round to nearest ...
    over = np.abs(h) > np.abs(x)                 # ... then step back toward zero where that rounded away from it
    h[over] = np.nextafter(h[over].astype(np.float16), np.float16(0)).astype(np.float64)
    assert (np.abs(h) <= np.abs(x)).all() and np.abs(h).max() < 2.0 ** 14
    hi = h / scale
    return hi, E.astype(np.float64) - hi, scale


@pytest.mark.parametrize("kind", ["uniform", "unit_rows", "tiny_and_big", "ints"])
def test_pair_bound_and_candidate_set(kind):
    rng = np.random.default_rng(hash(kind) % 2**31)
    n, dim, k = 4000, 256, 20
    if kind == "uniform":
        E = rng.uniform(-1, 1, (n, dim))
    elif kind == "unit_rows":
        E = rng.standard_normal((n, dim))
        E /= np.linalg.norm(E, axis=1, keepdims=True)
    elif kind == "tiny_and_big":  # rows 2^-9 of the largest: the low end of what the split arithmetic accepts (fp16 subnormals)
        E = rng.uniform(-1, 1, (n, dim)) * np.where(rng.random(n) < 0.5, 1.0, 2.0 ** -9)[:, None]
    else:
        E = rng.integers(-3, 4, (n, dim)).astype(np.float64)
    E = E.astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    hi, lo, _ = _split_hi(E)
    e_norm, lo_norm, q_norm = np.linalg.norm(E.astype(np.float64), axis=1), np.linalg.norm(lo, axis=1), float(np.linalg.norm(q.astype(np.float64)))
    assert (lo_norm <= 2.0 ** -10 * e_norm + 1e-30).all()  # the a-priori bound of a truncation to 11 bits ...
    exact = E.astype(np.float64) @ q.astype(np.float64)
    approx = hi @ q.astype(np.float64)
    # ... and the measured one the index keeps: per row |approx - exact| = |lo . q| <= |lo| |q|
    assert (np.abs(approx - exact) <= lo_norm * q_norm * (1 + 1e-12) + 1e-300).all()
    m = float(lo_norm.max()) * q_norm  # (the device adds 2^-12 |e| |q| for its fp32 roundings; float64 here needs none)
    kth = np.sort(approx)[::-1][k - 1]
    candidates = approx >= kth - 2 * m
    top_exact = np.argsort(-exact, kind="stable")[:k]
    assert candidates[top_exact].all()  # the exact top-k is inside the candidate set
    if kind != "ints":
        assert candidates.sum() < n // 4  # and the set is a real pruning, not everything


def test_maxsim_chunk_bound():
    """|approx - exact| of a chunk's MaxSim score <= max_j |e_lo,j| * sum_i |q_i| (max over rows, sum over query vectors)."""
    rng = np.random.default_rng(7)
    n, dim, nq = 3000, 128, 8
    E = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    sizes = rng.integers(1, 12, 600)
    off = np.concatenate(([0], np.cumsum(sizes)))
    off = off[off <= n]
    hi, lo, _ = _split_hi(E)
    S, A = E.astype(np.float64) @ Q.astype(np.float64).T, hi @ Q.astype(np.float64).T
    exact = np.array([S[off[c]:off[c + 1]].max(axis=0).sum() for c in range(len(off) - 1)])
    approx = np.array([A[off[c]:off[c + 1]].max(axis=0).sum() for c in range(len(off) - 1)])
    m = float(np.linalg.norm(lo, axis=1).max()) * float(np.linalg.norm(Q.astype(np.float64), axis=1).sum())
    assert (np.abs(approx - exact) <= m * (1 + 1e-12)).all()
    k = 10
    kth = np.sort(approx)[::-1][k - 1]
    assert (approx[np.argsort(-exact, kind="stable")[:k]] >= kth - 2 * m).all()


def _split_hi_rne(E: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """The rounding of the HI halves as the index builds them (since round 3): fp16(e * scale) rounded to NEAREST even."""
    mx = float(np.abs(E).max())
    scale = 2.0 ** (14 - int(np.floor(np.log2(mx)) + 1)) if mx > 0 else 1.0
    hi = (E.astype(np.float64) * scale).astype(np.float16).astype(np.float64) / scale
    return hi, E.astype(np.float64) - hi


def _query_hi(Q: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """What `query_planes_kernel` keeps of a query in its hi halves: fp16(q * q_scale), round to nearest, ONE power of two per
    query (largest |element| -> [2^13, 2^14)); q_lo = q - q_hi."""
    mx = float(np.abs(Q).max())
    q_scale = 2.0 ** (14 - int(np.floor(np.log2(mx)) + 1)) if mx > 0 else 1.0
    hi = (Q.astype(np.float64) * q_scale).astype(np.float16).astype(np.float64) / q_scale
    return hi, Q.astype(np.float64) - hi


@pytest.mark.parametrize("rne", [False, True])
def test_one_product_maxsim_bound(rne):
    """The one-product approximate pass (the default of the MaxSim batch; RAGLITE_HI_ONE_PRODUCT=0: two products): approx = sum_i max_j q_hi,i . e_hi,j.  Per pair
    s - a = q_lo . e + q_hi . e_lo, so |approx - exact| <= max|e_lo| sum_i |q_i| + (max|e| + max|e_lo|) sum_i |q_lo,i| -- the
    threshold `maxsim_threshold_kernel` computes when it is handed the queries' scales."""
    rng = np.random.default_rng(11 + rne)
    n, dim, nq, k = 3000, 128, 8, 10
    E = rng.uniform(-1, 1, (n, dim)).astype(np.float32)
    Q = rng.uniform(-1, 1, (nq, dim)).astype(np.float32)
    sizes = rng.integers(1, 12, 600)
    off = np.concatenate(([0], np.cumsum(sizes)))
    off = off[off <= n]
    e_hi, e_lo = _split_hi_rne(E) if rne else _split_hi(E)[:2]
    q_hi, q_lo = _query_hi(Q)
    S, A = E.astype(np.float64) @ Q.astype(np.float64).T, e_hi @ q_hi.T
    exact = np.array([S[off[c]:off[c + 1]].max(axis=0).sum() for c in range(len(off) - 1)])
    approx = np.array([A[off[c]:off[c + 1]].max(axis=0).sum() for c in range(len(off) - 1)])
    max_e, max_lo = float(np.linalg.norm(E.astype(np.float64), axis=1).max()), float(np.linalg.norm(e_lo, axis=1).max())
    q_norms, q_lo_norms = np.linalg.norm(Q.astype(np.float64), axis=1), np.linalg.norm(q_lo, axis=1)
    m = max_lo * float(q_norms.sum()) + (max_e + max_lo) * float(q_lo_norms.sum())
    assert (np.abs(approx - exact) <= m * (1 + 1e-12)).all()
    kth = np.sort(approx)[::-1][k - 1]
    cand = approx >= kth - 2 * m
    assert cand[np.argsort(-exact, kind="stable")[:k]].all()
    assert cand.sum() < len(exact) // 2  # still a pruning step
    # what rounding to nearest buys: the halves drop at most half an ulp, the bound's e_lo term roughly halves
    if rne:
        assert max_lo < 0.75 * float(np.linalg.norm(_split_hi(E)[1], axis=1).max())
    # and the queries' own term is the smaller one (round to nearest, 11 bits)
    assert (q_lo_norms <= 2.0 ** -11 * q_norms * (1 + 1e-9)).all()


@pytest.mark.parametrize("metric", ["cosine", "dot"])
@pytest.mark.parametrize("rne", [False, True])
def test_one_product_row_band_and_two_stage_candidates(metric, rne):
    """The batched half-bytes search (experimental, `api.hip: search_rows_fused_hi`): approximate similarity from q_hi . e_hi alone,
    band m of `row_threshold_kernel`; stage 1 keeps the rows reaching (k-th best of a row SAMPLE) - 2 m, stage 2 the rows reaching
    (k-th best approximate of those) - 2 m -- the exact top-k must survive both."""
    rng = np.random.default_rng(21 + rne + 2 * (metric == "dot"))
    n, dim, k, stride = 6000, 256, 20, 7
    E = rng.standard_normal((n, dim)).astype(np.float32)
    if metric == "dot":
        E *= rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    e_hi, e_lo = _split_hi_rne(E) if rne else _split_hi(E)[:2]
    q_hi, q_lo = _query_hi(q[None, :])
    q_hi, q_lo = q_hi[0], q_lo[0]
    E64, q64 = E.astype(np.float64), q.astype(np.float64)
    e_norm, lo_norm = np.linalg.norm(E64, axis=1), np.linalg.norm(e_lo, axis=1)
    qn, ql = float(np.linalg.norm(q64)), float(np.linalg.norm(q_lo))
    d_exact, d_approx = E64 @ q64, e_hi @ q_hi
    if metric == "cosine":
        exact, approx = d_exact / (e_norm * qn), d_approx / (e_norm * qn)
        ratio = float((lo_norm / e_norm).max())
        m = ratio + (1 + ratio) * ql / qn
    else:
        exact, approx = 1 + d_exact, 1 + d_approx
        m = float(lo_norm.max()) * qn + (float(e_norm.max()) + float(lo_norm.max())) * ql
    assert (np.abs(approx - exact) <= m * (1 + 1e-9)).all()
    top_exact = np.argsort(-exact, kind="stable")[:k]
    tau_s = np.sort(approx[::stride])[::-1][k - 1]           # k-th best approximate of the sample: <= the k-th best overall
    stage1 = np.flatnonzero(approx >= tau_s - 2 * m)
    assert np.isin(top_exact, stage1).all()
    a_k = np.sort(approx[stage1])[::-1][k - 1]
    assert a_k == np.sort(approx)[::-1][k - 1]               # the list's k-th entry IS the k-th best approximate overall
    stage2 = stage1[approx[stage1] >= a_k - 2 * m]
    assert np.isin(top_exact, stage2).all()
    assert len(stage2) < len(stage1) <= n // 2
