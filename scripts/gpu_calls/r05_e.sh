#!/bin/bash
# Round 5, call e: lazy images -- the whole GPU suite, then the bench line (index_memory of the default index).
set -u
TAG=${1:-r05_e}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | tail -40 | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
tail -c 1500 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
