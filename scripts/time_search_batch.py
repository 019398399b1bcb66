"""search_rows throughput at small batch sizes over 1 M x 1024 fp32 (cosine, exact top-100), with and without the half-bytes
search: python scripts/time_search_batch.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raglite_amd

n, d, k = 1_000_000, 1024, 100
E = torch.empty((n, d), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=2)
Q = torch.empty((64, d), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=20)
idx = raglite_amd.DeviceIndex(E, metric="cosine")
for B in (1, 4, 8, 16, 32):
    row = []
    for off in ("", "1"):
        idx.set_option("hi_search", 0 if off else 1)
        q = Q[:B] if B > 1 else Q[0]
        for _ in range(3):
            idx.search_rows(q, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            idx.search_rows(q, k)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 20)
    print(f"B = {B:3d}: half-bytes {row[0]:7.3f} ms ({B / row[0] * 1e3:8.0f} q/s)   full precision {row[1]:7.3f} ms ({B / row[1] * 1e3:8.0f} q/s)", flush=True)
idx.set_option("hi_search", 1)
