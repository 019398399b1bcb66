"""HBM read rate of the row-score stream pass (maxsim_stream_kernel, B = 1) against corpus size: is the ~6 TB/s of the 13 GB
pooling pass a property of the kernel or of the memory system at that footprint?  python scripts/stream_bw_vs_size.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raglite_amd

os.environ.setdefault("RAGLITE_NO_PLANES", "1")  # (no corpus image: only the streamed matrix is resident)
d = 1024
for n in (500_000, 1_000_000, 2_000_000, 3_200_000, 6_000_000):
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=1)
    q = torch.empty((1, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q, seed=2)
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    idx.time_kernel(1, q, 3)
    ms = idx.time_kernel(1, q, 10) / 10
    print(f"rows {n:>9}  {4.0 * n * d / 1e9:6.2f} GB  {ms:7.3f} ms  {4.0 * n * d / ms / 1e9:6.2f} TB/s", flush=True)
    idx.close()
    del E
