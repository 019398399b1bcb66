#!/bin/bash
# Round 5, call c: main-loop skeletons of three register tiles (scripts/micro/tile_shapes.hip) next to the kernel's own skeleton
# (experiment build, RAGLITE_PP_DBG=128: no block epilogues) and the full pass, same box.
set -u
OUT=gpurun_out/${1:-r05_c}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/tile_shapes.hip -o /tmp/tile_shapes 2> "$OUT/tile_shapes_build.err"; echo "build exit $?" | tee -a "$OUT/summary.txt"
timeout 300 /tmp/tile_shapes 2>&1 | tee -a "$OUT/summary.txt"
for d in 0 128 0 128; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  kernel, 8 passes per launch, DBG=$d: /" | cut -c1-170 | tee -a "$OUT/summary.txt"
done
for d in 0 128; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 16 2>/dev/null | tail -1 | sed "s/^/  kernel, one pass per launch, DBG=$d: /" | cut -c1-170 | tee -a "$OUT/summary.txt"
done
timeout 300 /tmp/tile_shapes 2>&1 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
