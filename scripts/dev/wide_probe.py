"""Kernel-level look at a MaxSim batch step over a wide index (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
nqr = int(sys.argv[3]) if len(sys.argv) > 3 else 64
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=5)
off = np.arange(0, n + 1, 8, dtype=np.int64)
Q = torch.empty((nqr, 32, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=50)
idx = raglite_amd.DeviceIndex(E, off, metric="dot")
for _ in range(2): idx.maxsim_topk_batch(Q, 100)
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(10): idx.maxsim_topk_batch(Q, 100)
torch.cuda.synchronize(); print("ms per step", (time.perf_counter() - t0) / 10 * 1e3, idx.filter_stats())
