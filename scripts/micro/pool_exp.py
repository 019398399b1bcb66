import sys, numpy as np, torch
sys.path.insert(0, ".")
import raglite_amd
from scripts.bench_configs import timed
raglite_amd.set_device(0)
d, S = 1024, 100_000
rng = np.random.default_rng(4)
lens = rng.integers(4, 61, size=S); ends = np.cumsum(lens); begins = ends - lens; T = int(ends[-1])
tokens = torch.empty((T, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(tokens, seed=4)
b = torch.as_tensor(begins, device="cuda"); e = torch.as_tensor(ends, device="cuda")
for rep in range(2):
    for norm in (True, False):
        ms = timed(lambda: raglite_amd.pool_norm(tokens, b, e, normalize=norm), 10)
        print("normalize", norm, round(ms, 4), "ms", round((4.0*T*d + 2.0*S*d)/ms/1e6, 1), "GB/s")
# long uniform spans: 32 rows each
lens2 = np.full(S, 32); ends2 = np.cumsum(lens2); begins2 = ends2 - lens2
b2 = torch.as_tensor(begins2, device="cuda"); e2 = torch.as_tensor(ends2, device="cuda")
ms = timed(lambda: raglite_amd.pool_norm(tokens[: int(ends2[-1])], b2, e2), 10)
print("uniform 32-row spans", round(ms, 4), "ms", round((4.0*int(ends2[-1])*d + 2.0*S*d)/ms/1e6, 1), "GB/s")
# pure streaming reference on the same buffer: the scan kernel
idx = raglite_amd.DeviceIndex(tokens, metric="dot")
q = torch.empty((1, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(q, seed=5)
ms = idx.time_kernel(1, q, 10) / 10
print("scan kernel over the same 13.1 GB", round(ms, 4), "ms", round(4.0*T*d/ms/1e6, 1), "GB/s")
