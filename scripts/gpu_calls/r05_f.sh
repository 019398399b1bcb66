#!/bin/bash
# Round 5, call f: after the test fixes for lazy images and the exact route over fp32-stored fp16 values -- the tests that failed / are new,
# the shaped corpora with fp16 queries, cfg 5 three times (a 39-ms iteration appeared once in the bench), then the whole suite.
set -u
TAG=${1:-r05_f}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_f16_exact.py tests/test_gpu_memory_budget.py "tests/test_gpu_hi_maxsim.py" tests/test_gpu_pp_pass.py -q --timeout 600 > "$OUT/pytest_subset.log" 2>&1; echo "subset exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_subset.log" | tail -10 | tee -a "$OUT/summary.txt"
for i in 1 2 3; do timeout 300 python scripts/bench_configs.py cfg5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_batch'], d['timing'], d['roofline'].get('kernel_ms'))" | tee -a "$OUT/summary.txt"; done
timeout 600 python scripts/bench_configs.py shaped_unit shaped_clustered > "$OUT/shaped.json" 2> "$OUT/shaped.err"; echo "shaped exit $?" | tee -a "$OUT/summary.txt"
python -c "
import json
for line in open('$OUT/shaped.json'):
    d = json.loads(line); print(d['workload'][-60:], d['value'], d['candidates_per_query'], d['f16_queries'])
" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | tail -10 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
