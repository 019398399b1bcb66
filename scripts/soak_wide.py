"""Randomised soak of round 6's wide-embedder routes (1024 < dim <= 4096, dim % 128 == 0) against float64 on the device:

    python scripts/soak_wide.py [seconds] [seed]

* `rl_maxsim_rerank` (maxsim_pairs_wide_kernel): random dim, nq 1..32, chunk layouts (ragged 1..15, one row, long chunks, empty chunks), padded
  candidate lists -- every score within 2e-6 of the score scale of float64, integer data exactly, -inf where there is nothing to score;
* `rl_maxsim_topk_batch` / `rl_maxsim_topk` over an index big enough for the HI image (score = sum_i max_{j in chunk} Q[i].D[j], the
  multi-vector form of `/root/reference/src/raglite/_search.py:143-149` behind the reranker call :394-396): batches of 1..20 queries, k 1..300,
  tie-aware comparison with float64, the route and its fallback flag recorded; integer data bit-exact against the full-precision passes;
* `rl_search_rows` (`_search.py:69-79`) of 1..4 queries (the packed scan over the HI plane) and of 96..200 queries (the fused top-k over the HI
  image), cosine / dot, also with thousands of copies of one row (the guarded full pass answers).
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

DIMS = [1152, 1280, 1536, 2048, 2560, 3072, 4096]
# fp32 sums of `dim` products against float64: relative to the scale the rounding acts on -- max |e| sum_i |q_i| for a MaxSim score, |e| |q| for a
# dot product, 1 for a cosine (north_star's "within 1e-4" on cosine-scale scores is 25 x this)
TOL = 4e-6


def scale(E, Q):
    return float(E.double().norm(dim=1).max() * Q.double().norm(dim=1).sum())


def offsets(rng, n, layout):
    if layout == 0:
        sizes = rng.integers(1, 16, size=n)
    elif layout == 1:
        return np.arange(n + 1, dtype=np.int64)
    elif layout == 2:
        sizes = np.concatenate(([min(n, 700)], rng.integers(1, 101, size=n)))
    else:
        sizes = rng.integers(0, 16, size=n)  # with empty chunks
    off = np.concatenate(([0], np.cumsum(sizes)))
    off = off[off <= n]
    if off[-1] != n:
        off = np.concatenate((off, [n]))
    return off.astype(np.int64)


def fill(shape, seed, integer):
    x = torch.empty(shape, dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(x, seed=seed)
    if integer:
        x = torch.round(x * 3.0)
    return x


def maxsim64(E, off, Q):
    """float64 MaxSim scores of every chunk on the device: [n_chunks]"""
    S = (E.double() @ Q.double().T)  # [n, nq]
    n_chunks = len(off) - 1
    seg = torch.repeat_interleave(torch.arange(n_chunks, device="cuda"), torch.as_tensor(np.diff(off), device="cuda"))
    out = torch.full((n_chunks, Q.shape[0]), float("-inf"), dtype=torch.float64, device="cuda")
    out.scatter_reduce_(0, seg[:, None].expand(-1, Q.shape[0]), S, reduce="amax")
    sc = out.sum(dim=1)
    sc[torch.as_tensor(np.diff(off) == 0, device="cuda")] = float("-inf")
    return sc


def check_topk(s, c, ref, k, tol, what):
    s, c = s.double(), c.long()
    kk = min(k, int((ref > float("-inf")).sum()))
    assert (c[kk:] == -1).all() and torch.isinf(s[kk:]).all(), what  # slots that cannot be filled: (-inf, -1)
    if kk == 0:
        return
    assert (c[:kk] >= 0).all(), what
    got = ref[c[:kk]]
    assert (s[:kk] - got).abs().max() <= tol, (what, float((s[:kk] - got).abs().max()), tol)
    assert (s[:kk][1:] <= s[:kk][:-1]).all(), what
    assert len(torch.unique(c[:kk])) == kk, what
    if kk:
        better = (ref > s[kk - 1] + 2 * tol).nonzero().flatten()
        missed = better[~torch.isin(better, c[:kk])]
        assert missed.numel() == 0, (what, missed[:5])


def rerank_case(rng, stats):
    d = int(rng.choice(DIMS))
    n = int(rng.integers(200, 6000))
    integer = bool(rng.integers(0, 3) == 0)
    off = offsets(rng, n, int(rng.integers(0, 4)))
    nq, nqr, nc = int(rng.integers(1, 33)), int(rng.integers(1, 9)), int(rng.choice([1, 7, 64, 200, 700]))
    E = fill((n, d), int(rng.integers(1, 1 << 30)), integer)
    Q = fill((nqr, nq, d), int(rng.integers(1, 1 << 30)), integer)
    cand = rng.integers(0, len(off) - 1, (nqr, nc)).astype(np.int32)
    cand[rng.random(cand.shape) < 0.05] = -1
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    got = idx.maxsim_rerank(Q, torch.as_tensor(cand, device="cuda"))
    got = got if torch.is_tensor(got) else torch.as_tensor(got, device="cuda")
    for b in range(nqr):
        ref = maxsim64(E, off, Q[b])
        want = torch.where(torch.as_tensor(cand[b] >= 0, device="cuda"), ref[torch.as_tensor(np.maximum(cand[b], 0), device="cuda").long()],
                           torch.tensor(float("-inf"), dtype=torch.float64, device="cuda"))
        fin = torch.isfinite(want)
        assert torch.isinf(got[b][~fin]).all()
        tol = 0.0 if integer else TOL * scale(E, Q[b])
        assert (got[b][fin].double() - want[fin]).abs().max() <= tol if fin.any() else True, ("rerank", d, n, nq, float((got[b][fin].double() - want[fin]).abs().max()), tol)
    idx.close()
    stats["rerank"] += 1


def maxsim_case(rng, stats):
    d = int(rng.choice(DIMS))
    n = int((64 << 20) // d + rng.integers(1000, 30_000))
    integer = bool(rng.integers(0, 3) == 0)
    f16 = bool(rng.integers(0, 3) == 0)  # fp16-STORED (the reference's storage precision)
    off = offsets(rng, n, int(rng.integers(0, 3)))  # (no empty chunks: the pass finds a chunk by counting chunk ends)
    nq, nqr, k = int(rng.integers(1, 33)), int(rng.choice([1, 2, 3, 8, 9, 17, 20])), int(rng.choice([1, 10, 100, 300]))
    E = fill((n, d), int(rng.integers(1, 1 << 30)), integer)
    if f16:
        E = E.half().float()
    Q = fill((nqr, nq, d), int(rng.integers(1, 1 << 30)), integer)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage="f16" if f16 else "f32")
    s, c = idx.maxsim_topk_batch(Q, k)
    st = idx.filter_stats()
    stats["maxsim_" + ("f16_" if f16 else "") + st["kind"] + ("_fallback" if st.get("fallback") else "")] += 1
    if bool(rng.integers(0, 3) == 0):  # one query under a metadata filter (+ tombstones)
        ok = rng.random(len(off) - 1) < float(rng.choice([0.5, 0.05, 0.0005]))
        dead = rng.choice(len(off) - 1, 5, replace=False)
        idx.delete_chunks(dead)
        sf, cf = idx.maxsim_topk(Q[0], k, chunk_filter=torch.as_tensor(ok, device="cuda"))
        ref = maxsim64(E, off, Q[0])
        live = torch.as_tensor(ok, device="cuda").clone()
        live[torch.as_tensor(dead, device="cuda")] = False
        ref[~live] = float("-inf")
        check_topk(sf, cf, ref, k, 1e-9 if integer else TOL * scale(E, Q[0]), ("maxsim filtered", d, n, nq, k, f16))
        stats["maxsim_filtered"] += 1
        idx.close()
        return
    for b in sorted({0, nqr - 1}):
        ref = maxsim64(E, off, Q[b])
        tol = 1e-9 if integer else TOL * scale(E, Q[b])
        check_topk(s[b], c[b], ref, k, tol, ("maxsim", d, n, nq, nqr, k))
    if nqr <= 2:
        s1, c1 = idx.maxsim_topk(Q[0], k)
        assert torch.equal(c1, c[0]) and torch.equal(s1.view(torch.int32), s[0].view(torch.int32))
    if integer:
        with idx.options(hi_maxsim=0):
            s0, c0 = idx.maxsim_topk_batch(Q, k)
        assert torch.equal(c0, c) and torch.equal(s0.view(torch.int32), s.view(torch.int32)), ("maxsim integer", d, n, nq, nqr, k)
    idx.close()


def rows_case(rng, stats):
    d = int(rng.choice(DIMS))
    n = int(max(66_000, (64 << 20) // d) + rng.integers(0, 40_000))
    metric = str(rng.choice(["cosine", "dot", "l2"]))
    B = int(rng.choice([1, 2, 3, 4, 5, 16, 96, 200]))
    k = int(rng.choice([1, 10, 100, 512]))
    f16 = bool(rng.integers(0, 3) == 0)
    E = fill((n, d), int(rng.integers(1, 1 << 30)), False)
    Q = fill((B, d), int(rng.integers(1, 1 << 30)), False)
    dup = bool(rng.integers(0, 5) == 0) and B <= 4
    if dup:
        rows = torch.as_tensor(rng.choice(n, 5000, replace=False), device="cuda")
        E[rows] = Q[0][None, :] * (1.0 if metric == "l2" else 0.9) + 1e-4 * torch.randn((5000, d), device="cuda")
    if f16:
        E = E.half().float()
    idx = raglite_amd.DeviceIndex(E, metric=metric, storage="f16" if f16 else "f32")
    masked = bool(rng.integers(0, 4) == 0)
    ok = torch.as_tensor(rng.random(n) < 0.4, device="cuda") if masked else None
    s, r = idx.search_rows(Q, k, chunk_filter=ok) if masked else idx.search_rows(Q, k)
    st = idx.filter_stats()
    stats["rows_" + metric + ("_f16_" if f16 else "_") + st["kind"] + ("_fallback" if st.get("fallback") else "")] += 1
    En = E.double().norm(dim=1)
    for b in sorted({0, B - 1}):
        if metric == "l2":
            ref = 1.0 - (E.double() - Q[b].double()[None, :]).norm(dim=1)
            tol = TOL * max(1.0, float(En.max() + Q[b].double().norm()))
        else:
            dots = E.double() @ Q[b].double()
            ref = dots / (En * Q[b].double().norm()) if metric == "cosine" else 1.0 + dots
            tol = TOL * (1.0 if metric == "cosine" else max(1.0, float(En.max() * Q[b].double().norm())))
        if masked:
            ref = torch.where(ok, ref, torch.tensor(float("-inf"), dtype=torch.float64, device="cuda"))
        check_topk(s[b], r[b], ref, k, tol, ("rows", metric, d, n, B, k, dup, f16, masked))
    idx.close()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    raglite_amd.set_device(0)
    rng = np.random.default_rng(seed)
    from collections import Counter

    stats = Counter()
    t0 = time.time()
    while time.time() - t0 < seconds:
        which = int(rng.integers(0, 10))
        if which < 4:
            rerank_case(rng, stats)
        elif which < 7:
            maxsim_case(rng, stats)
        else:
            rows_case(rng, stats)
        torch.cuda.empty_cache()
    print("soak_wide: all equal;", dict(stats), f"{time.time() - t0:.0f} s, seed {seed}")


if __name__ == "__main__":
    main()
