"""One MaxSim query per call over the headline corpus (the reranker's call pattern, `_search.py:394-396`): rl_maxsim_topk in a host loop.
    python scripts/time_one_query.py [n_queries] [name=value ...]      (route options as in bench.py --opt)
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split of the few-queries route."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
import raglite_amd  # noqa: E402

raglite_amd.set_device(0)
args = [a for a in sys.argv[1:] if "=" not in a]
for a in [a for a in sys.argv[1:] if "=" in a]:
    name, value = a.split("=", 1)
    raglite_amd.set_default_option(name, int(value))
n_q = int(args[0]) if args else 200
E = torch.empty((bench.N_ROWS, bench.DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=bench.SEED_CORPUS)
idx = raglite_amd.DeviceIndex(E, bench.chunk_offsets(bench.N_ROWS), metric="dot")
Q = torch.empty((n_q, bench.NQ, bench.DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=bench.SEED_QUERY)
for i in range(5):
    idx.maxsim_topk(Q[i], bench.TOPK)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n_q):
    out = idx.maxsim_topk(Q[i], bench.TOPK)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(json.dumps({"queries": n_q, "queries_per_s": n_q / t, "ms_per_query": 1e3 * t / n_q, "host_enqueue_ms_per_query": 1e3 * t_host / n_q,
                  "route": idx.filter_stats()["kind"]}))
