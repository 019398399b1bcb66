#!/bin/bash
# Round 4, GPU call AA: pool_norm_dma_kernel with the span's stores allowed for in its DMA wait (shipped) against without (experiment build,
# RAGLITE_POOL_BATCH=-2), same box; its parity tests first.
set -u
OUT=gpurun_out/${1:-r04_aa}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pool or embed or cfg4 or pooling" 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
for t in 0 -2 0 -2; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_POOL_BATCH=$t timeout 200 python scripts/time_pool.py "tune=$t" 2>/dev/null | tail -1 | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
