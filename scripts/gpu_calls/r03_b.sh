#!/bin/bash
# Round 3, GPU call B: the sixteen-queries-per-pass kernel (maxsim_pp.hip): parity, pass time next to the eight-query kernels, skeleton table, bench.
set -u
TAG=${1:-r03_b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest_pp.log"
timeout 300 python scripts/time_gemm_pass.py 1000000 20 6,7 > "$OUT/pass_times.json" 2> "$OUT/pass_times.err"; echo "pass times exit $?: $(cat "$OUT/pass_times.json")" | tee -a "$OUT/summary.txt"
for d in 1 2 10 11; do
  RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/pass_dbg$d.json" 2> /dev/null; echo "PP_DBG=$d: $(cat "$OUT/pass_dbg$d.json")" | tee -a "$OUT/summary.txt"
done
timeout 900 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_shaped.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "hi_maxsim or shaped or fullsize_maxsim" > "$OUT/pytest_pipeline.log" 2>&1
echo "pytest pipeline exit $?" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest_pipeline.log"
for nopp in 1 0; do
  RAGLITE_NO_PP=$nopp timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench_nopp$nopp.json" 2> "$OUT/bench_nopp$nopp.err"
  echo "bench NO_PP=$nopp exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT/bench_nopp$nopp.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.2f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms frac {r['roofline']['frac']:.3f} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
except Exception as exc:
    print("  (no bench line)", exc)
PY
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
