"""One cosine / l2 query at a time over a wide index (650 k x 1536): run under rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
metric = sys.argv[1] if len(sys.argv) > 1 else "cosine"
n, d = 650_000, 1536
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=2)
Q = torch.empty((64, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=20)
idx = raglite_amd.DeviceIndex(E, metric=metric)
for i in range(3): idx.search_rows(Q[i], 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): idx.search_rows(Q[i], 100)
torch.cuda.synchronize(); print(metric, "ms per query", (time.perf_counter() - t0) / 20 * 1e3, idx.filter_stats()["kind"])
