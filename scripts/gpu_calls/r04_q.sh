#!/bin/bash
# Round 4, GPU call Q: randomised soaks of what changed late in the round -- the batch pipeline (one launch for all passes, packed pairs, exact
# k-th threshold) and the B <= 16 row search / rerank -- against the oracle and the full-precision paths.
set -u
OUT=gpurun_out/${1:-r04_q}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 400 python scripts/soak_hi_batch.py 150 41 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/soak_pairs.py 90 42 2>&1 | tail -3 | tee -a "$OUT/summary.txt"
timeout 500 python scripts/soak_rows_hi.py 240 43 2>&1 | tail -5 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
