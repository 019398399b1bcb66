"""Kernel-only timing (HIP events on the launch stream) of the MaxSim pass kernels at the metric shape:
kind 2 = two queries per pass (maxsim_stream2_kernel), kind 3 = eight queries per pass over the pre-split
corpus image (maxsim_gemm_kernel, three fp16 products per multiply), kind 5 = eight queries over the HI image (two
products), kind 6 = the same with one product, kind 7 = SIXTEEN queries per pass over the HI image, one product
(maxsim_pp_kernel).
python scripts/time_gemm_pass.py [rows] [iters] [kinds, comma-separated] [queries given to the eight-query kernel] [queries given to
kind 7: a multiple of 16 -- more than 16 = that many / 16 passes in ONE launch, as the batch pipeline launches them]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raglite_amd  # noqa: E402
from bench import DIM, NQ, SEED_CORPUS, SEED_QUERY, chunk_offsets  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raglite_amd.set_device(0)
off = chunk_offsets(rows)
E = torch.empty((rows, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=SEED_CORPUS)
idx = raglite_amd.DeviceIndex(E, off, metric="dot")
NQ16 = int(sys.argv[5]) if len(sys.argv) > 5 else 16
Q = torch.empty((max(16, NQ16), NQ, DIM), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=SEED_QUERY)
out = {"rows": rows, "arithmetic": idx.arithmetic}
kinds = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3]
NQ8 = int(sys.argv[4]) if len(sys.argv) > 4 else 8  # queries given to the eight-query kernel (fewer: idle waves)
PRODUCTS = {2: 3, 3: 3, 5: 2, 6: 1, 7: 1}
for kind, nqueries in ((2, 2), (3, 8), (5, 8), (6, 8), (7, NQ16)):
    if kind not in kinds:
        continue
    qv = Q[:nqueries].reshape(nqueries * NQ, DIM)
    idx.time_kernel(kind, qv, 3)
    ms = idx.time_kernel(kind, qv, iters) / iters
    flops = PRODUCTS[kind] * 2.0 * nqueries * NQ * rows * DIM
    if kind == 7:
        ms = ms / (nqueries // 16)  # per pass of sixteen queries
        flops /= nqueries // 16
        nqueries = 16
    out[f"kind{kind}"] = {"ms_per_pass": ms, "queries_per_pass": nqueries, "queries_per_s": nqueries / ms * 1e3,
                          "ms_per_8_queries": ms * 8 / nqueries,
                          "hbm_GBs": (2.0 if kind in (5, 6, 7) else 4.0) * rows * DIM / ms / 1e6, "f16_mfma_TFs": flops / ms / 1e9}
print(json.dumps(out))
