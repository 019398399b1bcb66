"""Cost of the section-8f-1 lifecycle on one MI355X: filtered search vs unfiltered, delete, append.

    python scripts/bench_lifecycle.py    -> one JSON object
"""

from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import raglite_amd  # noqa: E402
from bench import chunk_offsets  # noqa: E402


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    raglite_amd.set_device(0)
    n, d = 1_000_000, 1024
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=2)
    off = chunk_offsets(n)
    n_chunks = len(off) - 1
    q = torch.empty((8, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q, seed=20)
    Q = torch.empty((32, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=21)
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    rng = np.random.default_rng(0)
    out = {"rows": n, "chunks": n_chunks}
    out["search_chunks_ms"] = timed(lambda: idx.search_chunks(q[0], 40, 10), 30)
    for dens in (0.5, 0.01):
        flt = torch.as_tensor(rng.random(n_chunks) < dens, device="cuda")
        bits = raglite_amd.pack_bits(flt)  # what a caller would cache per filter
        out[f"search_chunks_filtered_{dens}_ms"] = timed(lambda: idx.search_chunks(q[0], 40, 10, chunk_filter=bits), 30)
    out["maxsim_topk_ms"] = timed(lambda: idx.maxsim_topk(Q, 100), 30)
    flt = raglite_amd.pack_bits(rng.random(n_chunks) < 0.5)
    out["maxsim_topk_filtered_0.5_ms"] = timed(lambda: idx.maxsim_topk(Q, 100, chunk_filter=flt), 30)
    dead = rng.choice(n_chunks, 10_000, replace=False)
    t0 = time.perf_counter()
    idx.delete_chunks(dead)
    torch.cuda.synchronize()
    out["delete_10k_chunks_ms"] = (time.perf_counter() - t0) * 1e3
    out["search_chunks_after_delete_ms"] = timed(lambda: idx.search_chunks(q[0], 40, 10), 30)
    new = torch.empty((10_000, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(new, seed=22)
    sizes = np.full(1250, 8, dtype=np.int64)
    t0 = time.perf_counter()
    idx.append(new, sizes)  # first append: the borrowed 4.1 GB matrix is copied into owned, growable storage
    torch.cuda.synchronize()
    out["first_append_10k_rows_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    idx.append(new, sizes)
    torch.cuda.synchronize()
    out["second_append_10k_rows_ms"] = (time.perf_counter() - t0) * 1e3
    out["search_chunks_after_append_ms"] = timed(lambda: idx.search_chunks(q[0], 40, 10), 30)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
