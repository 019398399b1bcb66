"""Score error of the MaxSim stream kernel against float64 truth (run once per arithmetic mode).

    python scripts/split_error.py [normalized|uniform|tiny|wide] [--exact]
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import raglite_amd  # noqa: E402


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kind = args[0] if args else "normalized"
    rng = np.random.default_rng(5)
    n, d, nq = 40_000, 1024, 32
    E = rng.standard_normal((n, d))
    Q = rng.standard_normal((nq, d))
    if kind == "normalized":
        E /= np.linalg.norm(E, axis=1, keepdims=True)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    elif kind == "uniform":
        E, Q = rng.uniform(-1, 1, (n, d)), rng.uniform(-1, 1, (nq, d))
    elif kind == "tiny":  # a corpus whose elements span 12 orders of magnitude
        E *= 10.0 ** rng.uniform(-9, 3, (n, d))
        Q *= 10.0 ** rng.uniform(-6, 0, (nq, d))
    elif kind == "wide":
        E *= 3.0e4
        Q *= 1.0e-3
    E32, Q32 = E.astype(np.float32), Q.astype(np.float32)
    off = np.arange(n + 1, dtype=np.int64)  # one row per chunk: score = sum_i q_i . e
    idx = raglite_amd.DeviceIndex(E32, off, metric="dot")
    if "--exact" in sys.argv:
        idx.set_exact_fp32()
    got = np.asarray(idx.maxsim_scores(Q32), dtype=np.float64)
    truth = (E32.astype(np.float64) @ Q32.astype(np.float64).T).sum(axis=1)
    scale = (np.abs(E32.astype(np.float64)) @ np.abs(Q32.astype(np.float64)).T).sum(axis=1)  # sum |e||q|: the natural error unit
    err = np.abs(got - truth)
    print(json.dumps({"kind": kind, "arithmetic": idx.arithmetic, "max_abs_err": float(err.max()), "max_err_over_sum_abs": float((err / scale).max()),
                      "rms_err_over_sum_abs": float(np.sqrt(np.mean((err / scale) ** 2))), "max_abs_score": float(np.abs(truth).max())}))
    idx.close()


if __name__ == "__main__":
    main()
