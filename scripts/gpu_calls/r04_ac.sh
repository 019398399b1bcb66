#!/bin/bash
# Round 4, GPU call AC: a workgroup barrier in every other slab only (HALFBAR, experiment build RAGLITE_PP_DBG=8192): parity, then the pass time.
set -u
OUT=gpurun_out/${1:-r04_ac}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_HIP_LIB=$EXP RAGLITE_PP_DBG=8192 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
for d in 0 8192 0 8192; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  DBG=$d: /" | cut -c1-150 | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
