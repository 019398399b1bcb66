"""Randomised soak of the half-bytes row search for B <= 16 (api.hip: search_rows_hi -- candidates listed by the selection's final kernel,
guarded full pass riding on the re-scoring launch, one-launch guarded selection) and of the row-packing pairs kernel behind
rl_maxsim_rerank.

    python scripts/soak_rows_hi.py [seconds] [seed]

Row search (`/root/reference/src/raglite/_search.py:69-79`): integer data == the oracle bit for bit (ties to the lowest row); float data ==
the full-precision pass (option hi_search = 0) bit for bit; random B 1..16, k 1..512, cosine / dot, corpora with thousands of copies of the
best row (every list overflows: the guarded pass answers), metadata filters (the collecting flow), tombstones.
Rerank: packed == unpacked bit for bit, integer data == the oracle, over random dims / layouts / candidate lists with -1 pads."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import raglite_amd  # noqa: E402
from oracle import oracle  # noqa: E402


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def rows_case(rng):
    metric = "cosine" if rng.random() < 0.5 else "dot"
    dim = int(rng.choice([256, 512, 1024]))
    n = int((64 << 20) // dim + rng.integers(1, 20_000))  # just big enough for the HI plane
    B = int(rng.integers(1, 17))
    k = int(rng.choice([1, 7, 100, 300, 512]))
    kind = "small_int" if rng.random() < 0.5 else "uniform"
    E = oracle.synth_matrix(int(rng.integers(1, 1 << 30)), n, dim, kind)
    Q = oracle.synth_matrix(int(rng.integers(1, 1 << 30)), B, dim, kind)
    flavour = rng.integers(0, 4)
    if flavour == 1:  # thousands of copies of a row that wins for every query: lists overflow, massive ties
        hot = rng.choice(n, int(rng.integers(1100, 4000)), replace=False)
        E[hot] = (np.sign(Q.sum(axis=0)) * (2.0 if kind == "small_int" else 0.9))[None, :].astype(np.float32)
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    filt = None
    if flavour == 2:
        filt = rng.random(n) < rng.choice([0.001, 0.3, 0.9])
    if flavour == 3:
        idx.delete_chunks(rng.choice(n, n // 50, replace=False))
    S, R = idx.search_rows(Q, k, chunk_filter=filt)
    st = idx.filter_stats()
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(Q, k, chunk_filter=filt)
    assert np.array_equal(R, R0) and same(S, S0), ("hi != full", metric, dim, n, B, k, kind, int(flavour), st)
    if kind == "small_int" and filt is None and flavour != 3 and metric == "dot":
        for b in range(min(B, 2)):
            rs, rr = oracle.search_rows(E, Q[b], k, metric, np.float64)
            assert np.array_equal(R[b], rr), ("oracle rows", metric, dim, n, B, k, int(flavour))
    idx.close()
    return st["kind"], bool(st["fallback"])


def rerank_case(rng):
    dim = int(rng.choice([256, 384, 512, 768, 1024]))
    n = int(rng.choice([17, 300, 5000, 20000]))
    sizes = []
    tot = 0
    big = rng.random() < 0.3
    while tot < n:
        s = 0 if rng.random() < 0.05 else int(rng.integers(1, 300 if big else 16))
        s = min(s, n - tot)
        sizes.append(s)
        tot += s
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    nq = int(rng.integers(1, 33))
    nb = int(rng.choice([1, 2, 9, 128, 200]))
    n_cand = int(rng.choice([1, 5, 64, 100, 333]))
    E = oracle.synth_matrix(int(rng.integers(1, 1 << 30)), n, dim, "small_int")
    Q = np.stack([oracle.synth_matrix(int(rng.integers(1, 1 << 30)), nq, dim, "small_int") for _ in range(nb)])
    cand = rng.integers(-1, len(off) - 1, (nb, n_cand)).astype(np.int32)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    a = idx.maxsim_rerank(Q, cand)
    with idx.options(pairs_packed=0):
        b = idx.maxsim_rerank(Q, cand)
    assert same(a, b), ("packed != unpacked", dim, n, nq, nb, n_cand, big)
    ref = oracle.maxsim_scores(E, off, Q[0], np.float64)
    want = np.where(cand[0] >= 0, ref[np.maximum(cand[0], 0)], -np.inf)
    assert np.array_equal(a[0].astype(np.float64), want), ("oracle", dim, n, nq, nb, n_cand, big)
    idx.close()


def main() -> None:
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    cases = {"rows": 0, "rows_fallback": 0, "rerank": 0}
    while time.time() - t0 < budget:
        if rng.random() < 0.5:
            kind, fb = rows_case(rng)
            cases["rows"] += 1
            cases["rows_fallback"] += int(fb)
        else:
            rerank_case(rng)
            cases["rerank"] += 1
    print(f"soak_rows_hi: {cases} in {time.time() - t0:.0f} s, all equal")


if __name__ == "__main__":
    main()
