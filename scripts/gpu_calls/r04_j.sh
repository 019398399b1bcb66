#!/bin/bash
# Round 4, GPU call J: the slim index (rows + HI image, 1.5 x the corpus) -- its tests, the suites around what changed (ends bitmap, refresh order,
# option 19), and the headline bench with the default layout and with the slim one.
set -u
OUT=gpurun_out/${1:-r04_j}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_memory_budget.py tests/test_gpu_pp_pass.py tests/test_gpu_store.py tests/test_gpu_sharded.py -m gpu -x -q > "$OUT/pytest_a.log" 2>&1
echo "pytest A exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_a.log" | tee -a "$OUT/summary.txt"
for opts in "" "--opt keep_image=0 --opt keep_hi_plane=0"; do
  tag=$(echo "default $opts" | tr -d ' -' | tr '=' '_')
  timeout 600 python bench.py --steps 10 --warmup 3 $opts > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.err"
  echo "bench [$opts] exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT/bench_$tag.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  %.0f q/s  %.3f ms/step  pass %.4f ms frac %.3f cand %s fb %s" % (r["value"], r["ms_per_step"], r["roofline"].get("kernel_ms", float("nan")), r["roofline"]["frac"], r.get("candidates_per_query"), r.get("fallback_steps")))
print("  memory", r.get("index_memory"))
PY
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
