"""Kernel-level timing of the exact pairs re-scoring at the headline shape: 128 queries x 32 vectors x 1024, `n_cand` candidate chunks each
(ragged chunks of 1..15 rows of a 1 M-row corpus), through rl_maxsim_rerank (sanitise + maxsim_pairs[_packed]_kernel), HIP events.
python scripts/time_pairs.py [n_cand] [iters] [pairs_packed]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raglite_amd  # noqa: E402
from bench import DIM, NQ, SEED_CORPUS, SEED_QUERY, chunk_offsets  # noqa: E402

n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 100
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
packed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = 1_000_000
raglite_amd.set_device(0)
raglite_amd.set_default_option("keep_image", 0)
raglite_amd.set_default_option("keep_hi", 0)
raglite_amd.set_default_option("pairs_packed", packed)
off = chunk_offsets(rows)
E = torch.empty((rows, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=SEED_CORPUS)
idx = raglite_amd.DeviceIndex(E, off, metric="dot")
Q = torch.empty((128, NQ, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=SEED_QUERY)
g = torch.Generator(device="cuda").manual_seed(1)
cand = torch.randint(0, len(off) - 1, (128, n_cand), device="cuda", dtype=torch.int32, generator=g)
for _ in range(3):
    idx.maxsim_rerank(Q, cand)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    idx.maxsim_rerank(Q, cand)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
tiles_padded = 128 * n_cand
print(json.dumps({"n_cand": n_cand, "pairs_packed": packed, "ms_per_call": ms, "note": "sanitise kernel + pairs kernel",
                  "fp32_mfma_tflops_padded": 2.0 * tiles_padded * 16 * NQ * DIM / ms / 1e9}))
