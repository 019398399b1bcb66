"""Randomised soak of the bound-filtered MaxSim batch (sixteen-query approximate pass, candidate collection, exact re-scoring; DESIGN.md 4.1e /
4.2d) against the oracle, on indexes big enough to keep the image of the hi halves.

    python scripts/soak_hi_batch.py [seconds] [seed]

Integer data (exact in every arithmetic): the batch == the oracle's top-k, bit for bit, over random corpus sizes (66 k .. 140 k rows of 1024 / 512
dims), chunk layouts (ragged, one row per chunk, chunks of hundreds of rows), query lengths 1..32, batch sizes 3..40, k 1..300, fp32 and
fp16 storage, tombstones; the sharded form (three shards, one threshold) == the single index.
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
from oracle import oracle  # noqa: E402


def offsets(rng, n_rows):
    kind = int(rng.integers(0, 3))
    if kind == 0:
        sizes = rng.integers(1, 16, n_rows)
    elif kind == 1:
        sizes = np.ones(n_rows, np.int64)
    else:
        sizes = rng.integers(1, 400, n_rows // 100 + 2)
    off = np.concatenate(([0], np.cumsum(sizes)))
    off = off[off <= n_rows]
    if off[-1] != n_rows:
        off = np.append(off, n_rows)
    return off.astype(np.int64), kind


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    raglite_amd.set_device(0)
    t_end, cases, sharded, f16_routes = time.time() + seconds, 0, 0, 0
    while time.time() < t_end:
        dim = int(rng.choice([1024, 1024, 512]))
        n = int(rng.integers(66_000, 140_000)) * (1024 // dim)
        nq, B, k = int(rng.integers(1, 33)), int(rng.integers(3, 41)), int(rng.integers(1, 301))
        storage = "f16" if rng.random() < 0.35 else "f32"
        off, layout = offsets(rng, n)
        n_chunks = len(off) - 1
        k = min(k, n_chunks)
        E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
        raglite_amd.synth_fill(E, seed=int(rng.integers(1, 1 << 30)), kind="small_int")
        Q = torch.empty((B, nq, dim), dtype=torch.float32, device="cuda")
        raglite_amd.synth_fill(Q, seed=int(rng.integers(1, 1 << 30)), kind="small_int")
        idx = raglite_amd.DeviceIndex(E.half() if storage == "f16" else E, off, metric="dot", storage=storage)
        dead = None
        if rng.random() < 0.3:
            dead = np.sort(rng.choice(n_chunks, int(rng.integers(1, max(2, n_chunks // 50))), replace=False))
            idx.delete_chunks(dead)
        s, c = idx.maxsim_topk_batch(Q, k)
        st = idx.filter_stats()
        tag = f"case {cases}: n={n} dim={dim} nq={nq} B={B} k={k} {storage} layout={layout} dead={0 if dead is None else len(dead)} filter={st['kind']} fb={st['fallback']}"
        assert st["kind"] == "maxsim_batch_hi" or B % 8 in (1, 2), tag
        Eh, Qh = E.cpu().numpy(), Q.cpu().numpy()
        live = np.ones(n_chunks, bool)
        if dead is not None:
            live[dead] = False
        for b in sorted({0, B // 2, B - 1}):
            if dead is None:
                ws, wc = oracle.maxsim_topk(Eh, off, Qh[b], k, np.float32)
            else:
                ws, wc = oracle.maxsim_topk_filtered(Eh, off, Qh[b], k, live)
            gs, gc = s[b].cpu().numpy(), c[b].cpu().numpy()
            assert np.array_equal(gc[: len(wc)], wc), (tag, b, gc[:6], wc[:6])
            assert np.array_equal(gs[: len(ws)], np.asarray(ws, np.float32)), (tag, b)
        # round 5: the same queries handed over as fp16 values (rl_maxsim_topk_batch_f16): over fp16 values -- an fp16-stored corpus, or this
        # fp32-stored one whose integer entries ARE fp16 values -- the one-product pass is exact and its own top-k is returned: the same bits
        hs, hc = idx.maxsim_topk_batch(Q.half(), k)
        sth = idx.filter_stats()
        assert torch.equal(hc, c) and torch.equal(hs, s), ("fp16 queries", tag, sth)
        assert sth["kind"] in ("maxsim_batch_f16_exact", "maxsim_batch_hi") or B % 8 in (1, 2), (tag, sth)
        f16_routes += int(sth["kind"] == "maxsim_batch_f16_exact")
        if dead is None and storage == "f32" and n >= 3 * 66_000 * (1024 // dim) - 70_000 and rng.random() < 0.5 and B % 8 not in (1, 2):
            cuts = [0, n_chunks // 2, n_chunks]  # two shards, one threshold
            if all(int(off[hi] - off[lo]) * dim >= (64 << 20) for lo, hi in zip(cuts[:-1], cuts[1:])):
                shards = [raglite_amd.DeviceIndex(E[int(off[lo]) : int(off[hi])].clone(), off[lo : hi + 1] - off[lo], metric="dot") for lo, hi in zip(cuts[:-1], cuts[1:])]
                allg = torch.stack([sh.maxsim_batch_begin(Q, k) for sh in shards])
                ls, lc = [], []
                for r, sh in enumerate(shards):
                    a, b_ = sh.maxsim_batch_finish(Q, allg, r, k)
                    ls.append(a)
                    lc.append(torch.where(b_ >= 0, b_ + cuts[r], torch.full_like(b_, -1)))
                ms, mc = raglite_amd.merge_topk(torch.stack(ls), torch.stack(lc).to(torch.int32), k)
                assert torch.equal(ms, s) and torch.equal(mc.to(torch.int64), c.to(torch.int64)), ("sharded", tag)
                for sh in shards:
                    sh.close()
                sharded += 1
        idx.close()
        cases += 1
        if cases % 5 == 0:
            print(f"{cases} cases ({sharded} sharded) ok; last: {tag}", flush=True)
    print(f"soak_hi_batch: {cases} cases ({sharded} sharded, {f16_routes} fp16-query batches through the exact route), no failure, seed {seed}")


if __name__ == "__main__":
    main()
