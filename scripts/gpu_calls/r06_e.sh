#!/bin/bash
# Round 6, call e: the thread-maximum prefilter of the block top-k (tests; A/B on the headline, cfg 5, one query)
set -u
TAG=${1:-r06_e}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_topk_block.py tests/test_gpu_hi_few.py tests/test_gpu_fused_topk.py -m gpu -q --timeout 600 > "$OUT/pytest_new.log" 2>&1; echo "pytest new exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_new.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
for o in 2 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm --no-cpu-baseline --no-f16 --opt topk_block=$o > "$OUT/bench_block$o.json" 2> "$OUT/bench_block$o.err"; echo "bench topk_block=$o exit $?" | tee -a "$OUT/summary.txt"
  python scripts/bench_summary.py "$OUT/bench_block$o.json" | head -2 | tee -a "$OUT/summary.txt"
  timeout 300 python scripts/time_one_query.py 200 topk_block=$o | tee -a "$OUT/summary.txt"
  timeout 300 python scripts/bench_configs.py cfg5 topk_block=$o > "$OUT/cfg5_$o.json" 2> "$OUT/cfg5_$o.err"; python - "$OUT/cfg5_$o.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    print("  ", r["workload"][:40], {k: r.get(k) for k in ("value", "ms_per_batch", "ms_per_query")})
PY
done
trace() {  # name, mark, need, command...
  local name=$1 mark=$2 need=$3; shift 3
  rm -rf /tmp/tr_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" ); echo "$name exit $?" | tee -a "$OUT/summary.txt"
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" "$mark" $need > "$OUT/${name}_timeline.txt" 2>&1
  cat "$OUT/${name}_timeline.txt" | tee -a "$OUT/summary.txt"
}
trace headline query_planes_kernel maxsim_pp_kernel python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm
trace cfg5 query_rows_planes maxsim_pp_kernel python "$ROOT/scripts/bench_configs.py" cfg5
echo "== $(date) done" | tee -a "$OUT/summary.txt"
