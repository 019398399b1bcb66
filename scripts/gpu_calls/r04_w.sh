#!/bin/bash
# Round 4, GPU call W: after the last two edits (overflow-safe row total in the packed pairs kernel; the MaxSim batch's guarded selection in one
# launch): the suites around them, both soaks, a bench line.
set -u
OUT=gpurun_out/${1:-r04_w}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_pairs_packed.py tests/test_gpu_sharded.py tests/test_gpu_pp_pass.py tests/test_gpu_shaped.py -m gpu -x -q > "$OUT/pytest_a.log" 2>&1
echo "pytest A exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_a.log" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/soak_hi_batch.py 100 61 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/soak_rows_hi.py 100 62 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-f16 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("  %.0f q/s  %.3f ms/step  launch %.4f ms (%.4f per pass) frac %.3f cand %s fb %s" % (r["value"], r["ms_per_step"], rf["kernel_ms"], rf.get("kernel_ms_per_pass", float("nan")), rf["frac"], r.get("candidates_per_query"), r.get("fallback_steps")))
PY
echo "== $(date) done" | tee -a "$OUT/summary.txt"
