#!/bin/bash
# Round 6, call h: the headline tail with fewer launches (flag zeroed by the query-image kernel, the bound from its sums, list tails filled by
# exact_threshold_kernel, the fallback's host word written by guarded_select_kernel), query-image kernels on 1024 threads, cfg 5's folded
# launches.  Whole GPU suite, then timelines.
set -u
TAG=${1:-r06_h}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 --deselect tests/test_gpu_scale_2g.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_gpu.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm --no-cpu-baseline --no-f16 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python scripts/bench_summary.py "$OUT/bench.json" | head -2 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/time_one_query.py 200 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_configs.py cfg5 cfg2 > "$OUT/cfg52.json" 2> "$OUT/cfg52.err"; python - "$OUT/cfg52.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    print("  ", r["workload"][:40], {k: r.get(k) for k in ("value", "ms_per_batch", "ms_per_query")})
PY
trace() {  # name, mark, need, command...
  local name=$1 mark=$2 need=$3; shift 3
  rm -rf /tmp/tr_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" ); echo "$name exit $?" | tee -a "$OUT/summary.txt"
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" "$mark" $need > "$OUT/${name}_timeline.txt" 2>&1
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" x --tail 40 > "$OUT/${name}_tail.txt" 2>&1
  cat "$OUT/${name}_timeline.txt" | tee -a "$OUT/summary.txt"
}
trace headline query_planes_kernel maxsim_pp_kernel python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm
trace cfg5 query_rows_planes maxsim_pp_kernel python "$ROOT/scripts/bench_configs.py" cfg5
trace one maxsim_stream_kernel "" python "$ROOT/scripts/time_one_query.py" 100
echo "--- one tail" >> "$OUT/summary.txt"; cat "$OUT/one_tail.txt" >> "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
