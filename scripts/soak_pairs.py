"""Randomised soak of the batched MaxSim paths (pair kernel, fp32-split and fp16-stored) against the oracle.

    python scripts/soak_pairs.py [seconds] [seed]

Integer data (exact in every arithmetic): batch == one-at-a-time == oracle, bit for bit, over random dims, corpus sizes
(down to one row), chunk layouts (empty chunks, one giant chunk), query lengths 17..32, odd / even batch sizes, tombstones.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import raglite_amd  # noqa: E402
from oracle import oracle  # noqa: E402


def offsets(rng, n_rows):
    kind = rng.integers(0, 4)
    if kind == 0:  # ragged 1..15 with empties
        sizes = []
        tot = 0
        while tot < n_rows:
            s = 0 if rng.random() < 0.05 else int(rng.integers(1, 16))
            s = min(s, n_rows - tot)
            sizes.append(s)
            tot += s
    elif kind == 1:  # one row per chunk
        sizes = [1] * n_rows
    elif kind == 2:  # few big chunks (longer than a tile, crossing workgroup ranges)
        sizes = []
        tot = 0
        while tot < n_rows:
            s = min(int(rng.integers(1, 400)), n_rows - tot)
            sizes.append(s)
            tot += s
    else:  # a single chunk, plus trailing empties
        sizes = [n_rows, 0, 0]
    return np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)


def main() -> None:
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = 0
    while time.time() - t0 < budget:
        dim = int(rng.choice([128, 256, 384, 512, 768, 1024]))
        n_rows = int(rng.choice([1, 2, 15, 16, 17, 33, 100, 1000, 4099, 20000, 70000]))
        nq = int(rng.integers(17, 33))
        nb = int(rng.integers(2, 8))
        storage = "f16" if rng.random() < 0.4 else "f32"
        off = offsets(rng, n_rows)
        n_chunks = len(off) - 1
        E = oracle.synth_matrix(int(rng.integers(1, 1 << 30)), n_rows, dim, "small_int")
        Qb = np.stack([oracle.synth_matrix(int(rng.integers(1, 1 << 30)), nq, dim, "small_int") for _ in range(nb)])
        idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage=storage)
        k = int(min(n_chunks, rng.integers(1, 120)))
        dead = None
        if n_chunks > 4 and rng.random() < 0.4:
            dead = rng.choice(n_chunks, size=max(1, n_chunks // 5), replace=False)
            idx.delete_chunks(dead)
        bs, bc = idx.maxsim_topk_batch(Qb, k)
        for i in range(nb):
            ss, sc = idx.maxsim_topk(Qb[i], k)
            assert np.array_equal(bc[i], sc) and np.array_equal(bs[i], ss), ("batch != single", dim, n_rows, nq, nb, storage, i)
            if dead is None:
                ws, wc = oracle.maxsim_topk(E, off, Qb[i], k, np.float32)
                assert np.array_equal(bc[i][: len(wc)], wc) and np.array_equal(bs[i][: len(wc)], ws), ("oracle", dim, n_rows, nq, storage)
            else:
                live = bc[i][bc[i] >= 0]
                assert not np.isin(live, dead).any(), ("tombstone returned", dim, n_rows)
        idx.close()
        cases += 1
    print(f"soak OK: {cases} random cases in {time.time() - t0:.0f} s (seed {seed})")


if __name__ == "__main__":
    main()
