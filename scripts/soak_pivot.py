"""Randomised soak of round 6's selection shortcuts, route against route on the device (no CPU reference: every route must return the SAME BITS):

    python scripts/soak_pivot.py [seconds] [seed]

* `rl_topk` -- histogram / filter / final (topk_block = 0), the one-block route (1), the one-block route with the thread-maximum prefilter (2):
  random n <= 262 144, k 1..2048, normal / crowded / tied / constant / sorted / NaN-sprinkled scores, batches of 1..40 rows;
* the single-query row search (`ORDER BY dist LIMIT k`, `/root/reference/src/raglite/_search.py:69-79`) with candidates from a pivot over
  workgroup maxima (hi_pivot = 1) against the ranked route (hi_pivot = 0) and the full-precision pass (hi_search = 0): random n up to 1.5 M rows,
  dims 128..1024, B 1..16, k 1..512, cosine / dot / l2 (B <= 4), uniform and integer data (ties), thousands of copies of one row (the lists overflow: the guarded
  pass answers), quantised rows (massive ties at the k-th score);
* one or two MaxSim queries (`_search.py:143-149,394-396` generalised) through the pivot route against the ranked route (same bits) and the rows
  route (hi_few = 0: the same chunks; the same bits on integer data).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import raglite_amd  # noqa: E402


def same(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


def topk_case(rng):
    n = int(rng.choice([1, 3, 63, 1000, 4097, 46_336, 125_003, 262_144, int(rng.integers(1, 262_145))]))
    nq = int(rng.integers(1, 41)) if n < 60_000 else int(rng.integers(1, 6))
    k = int(rng.choice([1, 10, 100, 128, 129, 512, 513, 2048, int(rng.integers(1, 2049))]))
    kind = int(rng.integers(0, 7))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1, 1 << 30)))
    x = torch.randn((nq, n), device="cuda", generator=g)
    if kind == 1:
        x = 480.0 + 34.0 * x
    elif kind == 2:
        x = torch.randint(0, 3, (nq, n), device="cuda", generator=g).float()
    elif kind == 3:
        x = torch.ones((nq, n), device="cuda")
    elif kind == 4:
        x = torch.sort(x, dim=1, descending=bool(rng.integers(0, 2))).values.contiguous()
    elif kind == 5:
        x[torch.rand((nq, n), device="cuda", generator=g) < 0.3] = float("nan")
        x[torch.rand((nq, n), device="cuda", generator=g) < 0.1] = float("-inf")
    elif kind == 6:  # every large score in the 16-byte groups of a few threads: the prefilter overflows
        grp = (torch.arange(n, device="cuda") // 4) % 1024
        x[:, grp < 30] += 100.0
    out = []
    for route in (0, 1, 2):
        raglite_amd.set_default_option("topk_block", route)
        out.append(raglite_amd.topk(x, k))
    raglite_amd.set_default_option("topk_block", 2)
    for r in (1, 2):
        assert torch.equal(out[0][1], out[r][1]) and same(out[0][0], out[r][0]), ("topk", n, nq, k, kind, r)
    return f"topk n={n} nq={nq} k={k} kind={kind}"


def rows_case(rng):
    metric = str(rng.choice(["cosine", "dot", "l2"]))
    dim = int(rng.choice([128, 256, 512, 1024]))
    k = int(rng.choice([1, 5, 32, 100, 128, 160, 256, 400, 512]))  # (beyond 170: a maximum per wave of up to 512 workgroups, G <= 2048)
    lo = max((64 << 20) // dim + 1, 3 * k * 2048 if k <= 128 else 3 * k * 256)
    n = int(rng.integers(lo, max(lo + 1, min(1_500_000, (3 << 30) // (4 * dim)))))
    B = int(rng.integers(1, 5 if metric == "l2" else 17))
    kind = "small_int" if rng.random() < 0.4 else "uniform"
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=int(rng.integers(1, 1 << 30)), kind=kind)
    Q = torch.empty((B, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=int(rng.integers(1, 1 << 30)), kind=kind)
    flavour = int(rng.integers(0, 4))
    if flavour == 1:  # thousands of copies of a row that wins for every query
        hot = torch.randperm(n, device="cuda")[: int(rng.integers(1500, 4000))]
        E[hot] = (2.0 * torch.sign(Q.sum(dim=0)))[None, :]
    elif flavour == 2:  # quantised rows: massive ties
        E = (torch.round(E * 2.0) / 2.0).contiguous()
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    q = Q if B > 1 else Q[0]
    S, R = idx.search_rows(q, k)
    st = idx.filter_stats()
    with idx.options(hi_pivot=0):
        S1, R1 = idx.search_rows(q, k)
    with idx.options(hi_search=0):
        S0, R0 = idx.search_rows(q, k)
    ok = torch.equal(R, R1) and same(S, S1) and torch.equal(R, R0) and same(S, S0)
    assert ok, ("rows", metric, dim, n, B, k, kind, flavour, st)
    idx.close()
    del E
    return f"rows {metric} dim={dim} n={n} B={B} k={k} {kind} flavour={flavour} route={st['kind']} cand={st['candidates_per_query_max']} fb={st['fallback']}"


def few_case(rng):
    dim = int(rng.choice([256, 512, 1024]))
    k = int(rng.choice([1, 10, 64, 100, 128]))
    per = int(rng.choice([1, 2, 3]))  # rows per chunk (at most)
    n_chunks = max(3 * k * 256, (64 << 20) // dim // per + 1) + int(rng.integers(0, 30_000))
    sizes = rng.integers(1, per + 1, n_chunks)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    n = int(off[-1])
    if n * dim < (64 << 20):
        return "few skipped (too small)"
    kind = "small_int" if rng.random() < 0.5 else "uniform"
    nq = int(rng.integers(1, 33))
    nqueries = int(rng.integers(1, 3))
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=int(rng.integers(1, 1 << 30)), kind=kind)
    Q = torch.empty((nqueries, nq, dim), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=int(rng.integers(1, 1 << 30)), kind=kind)
    flavour = int(rng.integers(0, 3))
    if flavour == 1:
        hot = torch.randperm(n, device="cuda")[: int(rng.integers(2500, 5000))]
        E[hot] = (3.0 * Q[0].sum(dim=0))[None, :]
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    S, C = idx.maxsim_topk_batch(Q, k)
    st = idx.filter_stats()
    with idx.options(hi_pivot=0):
        S1, C1 = idx.maxsim_topk_batch(Q, k)
    with idx.options(hi_few=0):
        S0, C0 = idx.maxsim_topk_batch(Q, k)
    assert torch.equal(C, C1) and same(S, S1), ("few pivot vs ranked", dim, n, n_chunks, nqueries, nq, k, kind, flavour, st)
    if kind == "small_int" or st["fallback"]:
        assert torch.equal(C, C0) and same(S, S0), ("few vs rows", dim, n, n_chunks, nqueries, nq, k, kind, flavour, st)
    else:
        for b in range(nqueries):
            assert set(C[b].tolist()) == set(C0[b].tolist()), ("few vs rows (sets)", dim, n, n_chunks, nq, k, st)
    idx.close()
    del E
    return f"few dim={dim} chunks={n_chunks} rows={n} queries={nqueries}x{nq} k={k} {kind} flavour={flavour} route={st['kind']} cand={st['candidates_per_query_max']} fb={st['fallback']}"


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    raglite_amd.set_device(0)
    t0 = time.time()
    counts = {"topk": 0, "rows": 0, "few": 0}
    notes = {"rows_pivot": 0, "rows_fallback": 0, "few_fallback": 0}
    while time.time() - t0 < seconds:
        which = rng.choice(["topk", "topk", "rows", "few"])
        msg = {"topk": topk_case, "rows": rows_case, "few": few_case}[which](rng)
        counts[which] += 1
        if " fb=True" in msg:
            notes["rows_fallback" if which == "rows" else "few_fallback"] += 1
        if sum(counts.values()) % 25 == 0:
            print(f"[{time.time() - t0:6.0f} s] {counts} last: {msg}", flush=True)
        torch.cuda.empty_cache()
    print(f"soak_pivot OK: {counts} cases in {time.time() - t0:.0f} s, fallbacks {notes}, seed {seed}")


if __name__ == "__main__":
    main()
