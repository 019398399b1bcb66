"""One query at k = 256 (the reference's own num_hits: 4 x 64, `_search.py:66-67`, hybrid search) over 1 M x 1024: the pivot route against the ranked one, cosine and l2."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
n, d = 1_000_000, 1024
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=2)
Q = torch.empty((64, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=20)
for metric in ("cosine", "l2"):
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    for k in (100, 160, 256, 512):
        for piv in (1, 0):
            with idx.options(hi_pivot=piv):
                for i in range(3): idx.search_rows(Q[i], k)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(40): idx.search_rows(Q[i % 64], k)
                torch.cuda.synchronize()
                print(metric, "k", k, "hi_pivot", piv, "ms per query %.4f" % ((time.perf_counter() - t0) / 40 * 1e3), idx.filter_stats()["kind"], idx.filter_stats()["candidates_per_query_mean"], flush=True)
    idx.close()
