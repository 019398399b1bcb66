import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
def same(a, b): return torch.equal(a.view(torch.int32), b.view(torch.int32))
rng = np.random.default_rng(5)
found = 0
for trial in range(60):
    dim = int(rng.choice([128, 256])); k = int(rng.choice([1, 5, 32])); n = int(rng.integers(600_000, 900_000)); B = 1
    E = torch.empty((n, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=int(rng.integers(1, 1 << 30)))
    Q = torch.empty((B, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=int(rng.integers(1, 1 << 30)))
    flavour = int(rng.integers(0, 2))
    if flavour == 1:
        hot = torch.randperm(n, device="cuda")[: int(rng.integers(1500, 4000))]
        E[hot] = (2.0 * torch.sign(Q.sum(dim=0)))[None, :]
    idx = raglite_amd.DeviceIndex(E, metric="l2")
    S, R = idx.search_rows(Q[0], k); st = idx.filter_stats()
    with idx.options(hi_pivot=0): S1, R1 = idx.search_rows(Q[0], k)
    with idx.options(hi_search=0): S0, R0 = idx.search_rows(Q[0], k)
    ref = 1.0 - (E.double() - Q[0].double()[None, :]).norm(dim=1)
    top = torch.topk(ref, k)
    a = torch.equal(R, R1) and same(S, S1); b = torch.equal(R, R0) and same(S, S0)
    if not (a and b):
        found += 1
        print("MISMATCH dim", dim, "n", n, "k", k, "flavour", flavour, st["kind"], st["candidates_per_query_max"], st["fallback"], "vs pivot0", a, "vs full", b)
        print("  HI   rows", R.tolist(), [float(x) for x in S.tolist()])
        print("  piv0 rows", R1.tolist(), [float(x) for x in S1.tolist()])
        print("  full rows", R0.tolist(), [float(x) for x in S0.tolist()])
        print("  f64  rows", top.indices.tolist(), [float(x) for x in top.values.tolist()])
        if found >= 3: break
    idx.close(); del E
print("done, mismatches", found)
