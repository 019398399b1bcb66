#!/bin/bash
# Round 4, GPU call B: the test files call A did not reach (its -x stopped at a test whose own expectation was wrong), every other GPU
# test, smoke.
set -u
TAG=${1:-r04_b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/smoke.log"
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 -s > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -15 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
grep -ho "\[[a-z0-9_ =A-Z]*\] [^\[]*" "$OUT/pytest_gpu.log" | sort | uniq | head -60 >> "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
