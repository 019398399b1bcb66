#!/bin/bash
# Round 6, call i: the pivot route of the single-query row search (tests, A/B on cfg 2), the few-queries route without memsets, the whole suite.
set -u
TAG=${1:-r06_i}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_hi_pivot.py tests/test_gpu_hi_few.py tests/test_gpu_hi_search.py tests/test_gpu_hi_maxsim.py -m gpu -q --timeout 600 > "$OUT/pytest_new.log" 2>&1; echo "pytest new exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_new.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
for o in 1 0; do
  timeout 300 python scripts/bench_configs.py cfg2 hi_pivot=$o > "$OUT/cfg2_$o.json" 2> "$OUT/cfg2_$o.err"; python - "$OUT/cfg2_$o.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    print("  ", r["workload"][:40], {k: r.get(k) for k in ("value", "ms_per_batch", "ms_per_query")}, r["timing"], r["check"])
PY
done
timeout 300 python scripts/time_one_query.py 200 | tee -a "$OUT/summary.txt"
trace() {  # name, mark, need, command...
  local name=$1 mark=$2 need=$3; shift 3
  rm -rf /tmp/tr_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" ); echo "$name exit $?" | tee -a "$OUT/summary.txt"
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" x --tail $mark > "$OUT/${name}_tail.txt" 2>&1
  cat "$OUT/${name}_tail.txt" | tee -a "$OUT/summary.txt"
}
trace cfg2 24 "" python "$ROOT/scripts/cfg2_loop.py"
trace one 30 "" python "$ROOT/scripts/time_one_query.py" 100
( cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --memory-copy-trace --output-format csv -d /tmp/tr_hip -o t -- python "$ROOT/scripts/cfg2_loop.py" > /dev/null 2> "$OUT/hip.err" ); echo "hip trace exit $?" | tee -a "$OUT/summary.txt"
for f in $(find /tmp/tr_hip -name "*.csv"); do echo "--- $f"; head -3 "$f" | cut -c1-300; done >> "$OUT/summary.txt"
f=$(find /tmp/tr_hip -name "*memory_copy_trace.csv" | head -1); [ -n "$f" ] && tail -12 "$f" | cut -c1-300 >> "$OUT/summary.txt"
f=$(find /tmp/tr_hip -name "*hip_api_trace.csv" | head -1); [ -n "$f" ] && cut -d, -f3 "$f" | sort | uniq -c | sort -rn | head -20 >> "$OUT/summary.txt"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 --deselect tests/test_gpu_scale_2g.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_gpu.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
