"""Box-normalised timing of the headline kernel: the MaxSim stream kernel next to the single-query scan kernel.

    python scripts/kernel_ab.py [reps] [--f16] [--exact]

GPU boxes differ by a few percent (clocks under load), so kernel variants measured on different boxes are compared
through the ratio stream/scan: both kernels stream the same 4.096 GB corpus, and the scan kernel does not change
between variants.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import raglite_amd  # noqa: E402
from bench import chunk_offsets  # noqa: E402


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    f16 = "--f16" in sys.argv  # fp16-stored corpus: half the bytes per pass
    reps = int(args[0]) if args else 5
    n, d = 1_000_000, 1024
    raglite_amd.set_device(0)
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=1)
    Q = torch.empty((32, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=2)
    if f16:
        E = E.half()
    idx = raglite_amd.DeviceIndex(E, chunk_offsets(n), metric="dot", storage="f16" if f16 else "f32")
    if "--exact" in sys.argv:  # exact fp32 MFMA chain instead of the fp16 (hi, lo) split (include/raglite_hip.h)
        idx.set_exact_fp32()
    idx.time_kernel(0, Q, 5)
    idx.time_kernel(1, Q[:1], 5)
    Q2 = torch.empty((64, d), dtype=torch.float32, device="cuda")  # two queries of 32 vectors: the pair kernel
    raglite_amd.synth_fill(Q2, seed=3)
    stream, scan, pair = [], [], []
    try:
        idx.time_kernel(2, Q2, 3)
        has_pair = True
    except Exception:  # noqa: BLE001 - fp16-stored / exact-fp32 indexes have no pair kernel
        has_pair = False
    for _ in range(reps):
        stream.append(idx.time_kernel(0, Q, 20) / 20)
        scan.append(idx.time_kernel(1, Q[:1], 20) / 20)
        if has_pair:
            pair.append(idx.time_kernel(2, Q2, 20) / 20)
    gb = (2.0 if f16 else 4.0) * n * d / 1e9
    s, c = min(stream), min(scan)
    print(json.dumps({"storage": "f16" if f16 else "f32", "arithmetic": idx.arithmetic, "stream_ms": round(s, 4), "scan_ms": round(c, 4), "ratio": round(s / c, 4),
                      "stream_GBps": round(gb / s * 1e3, 1), "scan_GBps": round(gb / c * 1e3, 1),
                      "pair_ms": round(min(pair), 4) if pair else None, "pair_GBps": round(gb / min(pair) * 1e3, 1) if pair else None,
                      "stream_all": [round(x, 4) for x in stream], "scan_all": [round(x, 4) for x in scan],
                      "pair_all": [round(x, 4) for x in pair]}))


if __name__ == "__main__":
    main()
