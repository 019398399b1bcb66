import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, raglite_amd
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soak_wide import fill, offsets, maxsim64
rng = np.random.default_rng(3)
d, n, nq, k = 3072, 47747, 28, 300
for f16 in (True, False):
  for frac in (0.5, 0.05, 0.0005):
    off = offsets(rng, n, 0)
    E = fill((n, d), 5, False)
    if f16: E = E.half().float()
    Q = fill((1, nq, d), 6, False)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot", storage="f16" if f16 else "f32")
    ok = rng.random(len(off) - 1) < frac
    dead = rng.choice(len(off) - 1, 5, replace=False)
    idx.delete_chunks(dead)
    sf, cf = idx.maxsim_topk(Q[0], k, chunk_filter=torch.as_tensor(ok, device="cuda"))
    live = ok.copy(); live[dead] = False
    kk = min(k, int(live.sum()))
    cf_, sf_ = cf.cpu().numpy(), sf.cpu().numpy()
    print("f16", f16, "frac", frac, "live", int(live.sum()), "kk", kk, idx.filter_stats()["kind"], idx.filter_stats()["fallback"],
          "ids beyond kk != -1:", int((cf_[kk:] != -1).sum()), "scores beyond kk finite:", int(np.isfinite(sf_[kk:]).sum()),
          "masked ids returned:", int((~live[cf_[cf_ >= 0]]).sum()), "nan scores:", int(np.isnan(sf_).sum()))
    idx.close()
