"""Shared helpers for the parity tests (test infrastructure)."""

import numpy as np

F32 = np.float32


def sim_fp32_exact(E, q, metric):
    """fp32 as-computed similarity for INTEGER-valued data: dots / squared norms are exact in any
    accumulation order, the remaining ops (sqrt, mul, div, sub) are single IEEE fp32 operations in the
    same order as raglite_amd/csrc/scan.hip:finish_score -> bit-identical to the GPU."""
    E64, q64 = E.astype(np.float64), q.astype(np.float64)
    if metric == "l2":
        d2 = ((E64 - q64[None, :]) ** 2).sum(axis=1).astype(F32)
        return F32(1.0) - np.sqrt(d2)
    dot = (E64 @ q64).astype(F32)
    if metric == "dot":
        return F32(1.0) + dot
    ne = np.sqrt((E64 * E64).sum(axis=1).astype(F32))
    nq = np.sqrt(F32((q64 * q64).sum()))
    with np.errstate(invalid="ignore", divide="ignore"):
        c = dot / (ne * nq)
    return F32(1.0) - (F32(1.0) - c)


def assert_topk_close(gpu_scores, gpu_ids, all_scores64, k, tol):
    """Tie-aware top-k comparison against exact fp64 scores of EVERY element.

    (1) each returned score matches the oracle's score of the returned id within tol;
    (2) returned scores are non-increasing and ids are distinct;
    (3) nothing clearly better was missed: every element whose oracle score exceeds the k-th returned
        score by more than 2*tol is among the returned ids."""
    gpu_scores = np.asarray(gpu_scores, dtype=np.float64)
    gpu_ids = np.asarray(gpu_ids, dtype=np.int64)
    n = len(all_scores64)
    kk = min(k, n)
    assert np.all(gpu_ids[:kk] >= 0) and np.all(gpu_ids[kk:] == -1)
    assert len(set(gpu_ids[:kk].tolist())) == kk
    got = all_scores64[gpu_ids[:kk]]
    np.testing.assert_allclose(gpu_scores[:kk], got, rtol=0, atol=tol)
    assert np.all(np.diff(gpu_scores[:kk]) <= 0)
    if kk:
        missed = np.setdiff1d(np.nonzero(all_scores64 > gpu_scores[kk - 1] + 2 * tol)[0], gpu_ids[:kk])
        assert missed.size == 0, f"missed {missed[:5]}"
    assert np.all(np.isneginf(gpu_scores[kk:]))


def ragged_offsets(rng, n_rows, lo=1, hi=15, empty_every=0):
    sizes = []
    total = 0
    while total < n_rows:
        s = int(rng.integers(lo, hi + 1))
        if empty_every and len(sizes) % empty_every == empty_every - 1:
            s = 0
        s = min(s, n_rows - total)
        sizes.append(s)
        total += s
    return np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
