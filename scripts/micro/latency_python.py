"""End-to-end latency of the drop-in `vector_search(np.ndarray)` (NumPy in, Python lists out) on small indexes."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import raglite_amd
raglite_amd.set_device(0)
out = {}
rng = np.random.default_rng(0)
for n_chunks in (2_000, 20_000, 200_000):
    d = 1024
    E = rng.standard_normal((n_chunks * 5, d)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    off = np.arange(0, n_chunks * 5 + 1, 5, dtype=np.int64)
    gi = raglite_amd.GpuIndex([f"{i:016x}" for i in range(n_chunks)], E, chunk_offsets=off)
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    q = E[123].astype(np.float16)
    for _ in range(5):
        raglite_amd.vector_search(q, num_results=10, config=cfg, index=gi)
    t0 = time.perf_counter()
    for _ in range(300):
        ids, sc = raglite_amd.vector_search(q, num_results=10, config=cfg, index=gi)
    out[f"vector_search_{n_chunks * 5}_rows_us"] = round((time.perf_counter() - t0) / 300 * 1e6, 1)
    gi.close()
print(json.dumps(out))
