import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raglite_amd
from oracle import oracle
rng = np.random.default_rng(50)
n, dim, k = 140_000, 512, 100
off = np.arange(n + 1, dtype=np.int64)
E = oracle.synth_matrix(9870, n, dim)
Q = oracle.synth_matrix(9871, 8, dim)
hot = rng.choice(n, 4000, replace=False)
E[hot] = (3.0 * Q.sum(axis=0)[None, :] + 1e-4 * rng.standard_normal((4000, dim))).astype(np.float32)
for pv in (1, 0):
    raglite_amd.set_default_option("hi_pivot", pv)
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    s, c = idx.maxsim_topk(Q, k)
    print("pivot", pv, idx.filter_stats(), s[:3], c[:3], np.isin(c, hot).all())
    with idx.options(hi_few=0):
        fs, fc = idx.maxsim_topk(Q, k)
    print("   same as hi_few=0:", np.array_equal(c, fc), np.array_equal(s.view(np.uint32), fs.view(np.uint32)))
    idx.close()
