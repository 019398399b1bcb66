#!/bin/bash
# Round 4, GPU call AE: soaks on the round's final code -- the fused row top-k of big batches (MODE 2 of the pass kernel), the pair paths.
set -u
OUT=gpurun_out/${1:-r04_ae}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 400 python scripts/soak_fused_rows.py 180 71 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 200 python scripts/soak_pairs.py 60 72 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
