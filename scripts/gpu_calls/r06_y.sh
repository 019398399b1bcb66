#!/bin/bash
# Round 6, call y: soaks on the round's code -- the new selection shortcuts route against route (scripts/soak_pivot.py), the bound-filtered MaxSim
# batch and the row search against the oracle (soak_hi_batch.py, soak_rows_hi.py), the fused row top-k (soak_fused_rows.py).
set -u
TAG=${1:-r06_y}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 700 python scripts/soak_pivot.py 300 5 > "$OUT/soak_pivot.txt" 2>&1; echo "soak_pivot exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/soak_pivot.txt" | cut -c1-400 | tee -a "$OUT/summary.txt"
timeout 500 python scripts/soak_hi_batch.py 200 71 > "$OUT/soak_hi_batch.txt" 2>&1; echo "soak_hi_batch exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/soak_hi_batch.txt" | cut -c1-400 | tee -a "$OUT/summary.txt"
timeout 500 python scripts/soak_rows_hi.py 200 72 > "$OUT/soak_rows_hi.txt" 2>&1; echo "soak_rows_hi exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/soak_rows_hi.txt" | cut -c1-400 | tee -a "$OUT/summary.txt"
timeout 500 python scripts/soak_fused_rows.py 200 73 > "$OUT/soak_fused_rows.txt" 2>&1; echo "soak_fused_rows exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/soak_fused_rows.txt" | cut -c1-400 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
