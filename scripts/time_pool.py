"""Times rl_pool_norm on the BASELINE cfg 4 shape (3.2 M x 1024 token rows -> 100 k spans, fp16 out) -- for A/B runs under
RAGLITE_POOL_VGPR=1 / RAGLITE_POOL_BATCH.  python scripts/time_pool.py [tag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raglite_amd

rng = np.random.default_rng(4)
sizes = rng.integers(4, 61, 100_000)
b = np.concatenate(([0], np.cumsum(sizes[:-1]))).astype(np.int64)
e = b + sizes
T, d = int(e[-1]), 1024
tokens = torch.empty((T, d), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(tokens, seed=4)
bd, ed = torch.as_tensor(b, device="cuda"), torch.as_tensor(e, device="cuda")
for _ in range(3):
    raglite_amd.pool_norm(tokens, bd, ed)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    raglite_amd.pool_norm(tokens, bd, ed)
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 10
gb = (4.0 * T * d + 2.0 * len(b) * d) / 1e9
print(f"{sys.argv[1] if len(sys.argv) > 1 else '':>28}  {ms:7.3f} ms  {gb / ms:6.3f} TB/s  ({T} rows)")
