"""Latency of a single-query search on small indexes (the reference's typical database sizes)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import raglite_amd
raglite_amd.set_device(0)
out = {}
for n in (10_000, 100_000, 1_000_000):
    d = 1024
    E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=1)
    off = np.arange(0, n + 1, 5, dtype=np.int64)
    idx = raglite_amd.DeviceIndex(E, off, metric="cosine")
    q = torch.empty((d,), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(q, seed=2)
    qh = q.cpu().numpy()
    for name, fn in (("search_rows_dev", lambda: idx.search_rows(q, 10)), ("search_chunks_dev", lambda: idx.search_chunks(q, 40, 10)),
                     ("search_chunks_host_args", lambda: idx.search_chunks(qh, 40, 10))):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); out[f"{name}_n{n}_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    idx.close(); del E
print(json.dumps(out))
