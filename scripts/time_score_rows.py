"""Kernel-only timing of the batched similarity (rl_search_rows' scoring stage, no selection): cfg 5's shard shape.
python scripts/time_score_rows.py [rows] [B] [metric ...]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raglite_amd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
metrics = sys.argv[3:] or ["cosine", "dot"]
raglite_amd.set_device(0)
E = torch.empty((rows, 1024), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(E, seed=5)
Q = torch.empty((B, 1024), dtype=torch.float32, device="cuda")
raglite_amd.synth_fill(Q, seed=50)
out = {"rows": rows, "B": B}
for metric in metrics:
    idx = raglite_amd.DeviceIndex(E, metric=metric)
    idx.time_kernel(1, Q, 1)
    ms = idx.time_kernel(1, Q, 3) / 3
    out[metric] = {"ms": ms, "f16_mfma_TFs": 3 * 2.0 * B * rows * 1024 / ms / 1e9}
    idx.close()
print(json.dumps(out))
