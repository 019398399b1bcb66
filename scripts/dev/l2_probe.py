import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, raglite_amd
n, d, k = 1_000_000, 1024, 100
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=2)
q = torch.empty((64, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(q, seed=20)
for metric in ("l2", "cosine"):
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    with idx.options(hi_search=0):
        for i in range(12): idx.search_rows(q[i], k)
    torch.cuda.synchronize()
    idx.close()
