#!/bin/bash
# Round 4, GPU call R: the query fragments straight to registers (QREG; experiment build, RAGLITE_PP_QREG=1): parity tests of the pass through
# it, then the pass time against the shipped kernel on the same box (one pass per launch and the 8-pass launch; 1 M rows and a 1/8 shard).
set -u
OUT=gpurun_out/${1:-r04_r}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=1 timeout 900 python -m pytest tests/test_gpu_pp_pass.py -m gpu -x -q > "$OUT/pytest_qreg.log" 2>&1
echo "pytest (QREG) exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest_qreg.log" | tee -a "$OUT/summary.txt"
for rows in 1000000 125000; do for nq16 in 16 128; do
  for q in 0 1; do
    RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=$q timeout 300 python scripts/time_gemm_pass.py $rows 20 7 8 $nq16 2>/dev/null | tail -1 | sed "s/^/  rows $rows, $nq16 queries per launch, QREG=$q: /" | cut -c1-200 | tee -a "$OUT/summary.txt"
  done
done; done
for q in 0 1 0 1; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=$q RAGLITE_PP_DBG=128 timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  no epilogue, QREG=$q: /" | cut -c1-160 | tee -a "$OUT/summary.txt"
done
for q in 0 1 0 1; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=$q RAGLITE_PP_DBG=256 timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  no stores, QREG=$q: /" | cut -c1-160 | tee -a "$OUT/summary.txt"
done
for q in 0 1 0 1; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=$q timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  full, QREG=$q: /" | cut -c1-160 | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
