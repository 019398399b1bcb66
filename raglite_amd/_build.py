"""Build libraglite_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

`python -m raglite_amd._build` or `raglite_amd._build.build()`.  The library lands in
`raglite_amd/_lib/` (git-ignored, shipped to the GPU box by gpurun).  Objects are rebuilt only when a
source or header is newer, so repeated calls are cheap.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB_DIR = PKG / "_lib"
LIB_PATH = LIB_DIR / "libraglite_hip.so"
SOURCES = ["api.hip", "synth.hip", "pool_norm.hip", "scan.hip", "select.hip", "maxsim_stream.hip", "maxsim_generic.hip",
           "maxsim_gemm.hip", "maxsim_pp.hip", "score_gemm.hip", "mask.hip", "scan16.hip", "adapter_fit.hip", "partition_sim.hip", "comm.hip", "hi_filter.hip"]
# No -ffast-math: parity relies on IEEE fp32 divide / sqrt and on un-fused, un-reassociated sums.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


EXP_LIB_PATH = LIB_DIR / "libraglite_hip_exp.so"


def build(force: bool = False, verbose: bool = False, experiments: bool = False) -> Path:
    """`experiments=True` builds a SECOND library, libraglite_hip_exp.so, with -DRAGLITE_EXPERIMENTS: the timing-skeleton / trace
    instantiations of the kernels (some return wrong results by design) and their RAGLITE_* environment switches exist only there.
    scripts/gpu_calls/ load it through RAGLITE_HIP_LIB; nothing else does."""
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / ("obj_exp" if experiments else "obj")
    obj_dir.mkdir(exist_ok=True)
    lib_path = EXP_LIB_PATH if experiments else LIB_PATH
    flags = FLAGS + (["-DRAGLITE_EXPERIMENTS"] if experiments else [])
    headers = [CSRC / "common.h", INCLUDE / "raglite_hip.h"]
    hipcc = _hipcc()

    def compile_one(src: str) -> Path:
        obj = obj_dir / (src + ".o")
        if force or _newer(obj, [CSRC / src, *headers]):
            cmd = [hipcc, *flags, "-c", str(CSRC / src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    if force or _newer(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(lib_path)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stderr}")
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
