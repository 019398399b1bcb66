import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import raglite_amd
raglite_amd.set_device(0)
for a in sys.argv[1:]:
    n_, v_ = a.split('='); raglite_amd.set_default_option(n_, int(v_))
n, d, B, k = 1_250_000, 1024, 1000, 100
E = torch.empty((n, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=5)
Q = torch.empty((B, d), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=50)
idx = raglite_amd.DeviceIndex(E, metric="cosine")
print("mem after create", {k_: v for k_, v in idx.memory().items() if k_ in ("presplit_image","hi_image","hi_plane")})
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.search_rows(Q, k)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("per call ms:", " ".join(f"{t:.2f}" for t in ts))
print("mem after", {k_: v for k_, v in idx.memory().items() if k_ in ("presplit_image","hi_image","hi_plane","scratch")})
