"""MaxSim batch over an index with more chunks than the one-block selection takes (> 262 144): how long does the selection of the approximate
scores take when they crowd into one bin?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
n, d = 2_400_000, 1024
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=3)
off = np.arange(0, n + 1, 8, dtype=np.int64)
Q = torch.empty((128, 32, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=30)
idx = raglite_amd.DeviceIndex(E, off, metric="dot")
for _ in range(2): idx.maxsim_topk_batch(Q, 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): idx.maxsim_topk_batch(Q, 100)
torch.cuda.synchronize(); print("ms per 128-query step", (time.perf_counter() - t0) / 5 * 1e3, idx.filter_stats())
