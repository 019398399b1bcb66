#!/bin/bash
# Round 5, call b: the fp16-queries route (tests), the bench line with its new blocks (sustained MFMA rate, cfg 1, f16_queries,
# fraction check), and the rest of the GPU suite.
set -u
TAG=${1:-r05_b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_f16_exact.py -q -x --timeout 600 > "$OUT/pytest_f16_exact.log" 2>&1; echo "f16_exact tests exit $?" | tee -a "$OUT/summary.txt"
tail -15 "$OUT/pytest_f16_exact.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 --deselect tests/test_gpu_f16_exact.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/pytest_gpu.log"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
