"""The vendor's fp16 GEMM (hipBLASLt / rocBLAS through torch.mm) at the pass kernel's shapes, on its own -- run under
`rocprofv3 --kernel-trace --stats` to learn the NAME of the kernel the library picks (a Tensile name encodes macro-tile, wave grid,
LDS / direct-to-LDS, prefetch depth), next to `bench.py`'s `roofline.vendor_gemm_*` figures of the same box.

    python scripts/vendor_gemm.py [rows]        -> one JSON object (bench.vendor_gemm_calibration's) on stdout
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
import raglite_amd  # noqa: E402

raglite_amd.set_device(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else bench.N_ROWS
print(json.dumps(bench.vendor_gemm_calibration(torch.device("cuda", 0), rows, 20)))
