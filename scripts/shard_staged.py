"""What one rank of an N-way sharded MaxSim batch does per step with ONE candidate threshold for all shards (rl_maxsim_batch_begin /
_finish), measured on ONE GPU: the headline corpus cut into N shards, every shard's approximate lists computed once, then shard 0's two
halves timed (the all-gather between them is not: (N, 128, 101) float32).  python scripts/shard_staged.py [N ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raglite_amd
from bench import DIM, N_ROWS, NQ, SEED_CHUNKS, SEED_CORPUS, SEED_QUERY, TOPK, chunk_offsets
from raglite_amd._sharded import shard_bounds_by_chunk

raglite_amd.set_device(0)
B = 128
off = chunk_offsets(N_ROWS)
E = torch.empty((N_ROWS, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=SEED_CORPUS)
Q = torch.empty((B, NQ, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=SEED_QUERY)


def timed(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for world in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    shards = []
    for lo, hi in shard_bounds_by_chunk(off, world):
        r0, r1 = int(off[lo]), int(off[hi])
        shards.append(raglite_amd.DeviceIndex(E[r0:r1], off[lo : hi + 1] - off[lo], metric="dot"))
    alone = timed(lambda: shards[0].maxsim_topk_batch(Q, TOPK))
    cand_alone = shards[0].filter_stats()["candidates_per_query_mean"]
    allg = torch.stack([sh.maxsim_batch_begin(Q, TOPK) for sh in shards])
    t_begin = timed(lambda: shards[0].maxsim_batch_begin(Q, TOPK))
    shards[0].maxsim_batch_begin(Q, TOPK)

    def fin():
        shards[0].maxsim_batch_begin(Q, TOPK)
        shards[0].maxsim_batch_finish(Q, allg, 0, TOPK)

    t_both = timed(fin)
    st = shards[0].filter_stats()
    print(json.dumps({"world": world, "rows_per_shard": int(shards[0].n_rows), "ms_alone": round(alone, 3), "candidates_alone": round(cand_alone, 1),
                      "ms_begin": round(t_begin, 3), "ms_begin_plus_finish": round(t_both, 3), "candidates_staged": round(st["candidates_per_query_mean"], 1),
                      "fallback": st["fallback"], "queries_per_s_upper_bound": round(B / t_both * 1e3), "queries_per_s_alone": round(B / alone * 1e3)}))
    for sh in shards:
        sh.close()
