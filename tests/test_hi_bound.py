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
