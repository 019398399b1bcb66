#!/bin/bash
# Round 3, GPU call A: everything round 2 left "not yet run" + the new bench blocks + the cfg 5 timing the driver could not reproduce.
set -u
TAG=${1:-r03_a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_hi_search.py tests/test_gpu_fused_topk.py -m gpu -q --timeout 600 > "$OUT/pytest_experimental.log" 2>&1
echo "pytest experimental exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_experimental.log"
timeout 900 python -m pytest tests/test_gpu_shaped.py tests/test_gpu_sharded.py -m gpu -q --timeout 600 > "$OUT/pytest_new.log" 2>&1
echo "pytest new exit $?" | tee -a "$OUT/summary.txt"; tail -30 "$OUT/pytest_new.log"
RAGLITE_GEMM_DEEP=1 timeout 600 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_fullsize.py -m gpu -q -x -k "hi_maxsim or fullsize_maxsim" --timeout 600 > "$OUT/pytest_deep.log" 2>&1
echo "pytest deep exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_deep.log"
gcc -O2 -std=c11 -Iinclude scripts/micro/r3_probe.c -o /tmp/r3_probe -Lraglite_amd/_lib -lraglite_hip -lm -Wl,-rpath,$PWD/raglite_amd/_lib
for deep in 0 1; do
  echo "== RAGLITE_GEMM_DEEP=$deep" | tee -a "$OUT/summary.txt"
  RAGLITE_GEMM_DEEP=$deep R3_PROBE_DEBUG=1 timeout 120 /tmp/r3_probe 2>&1 | tee "$OUT/probe_deep$deep.txt" | grep -E "pass kind|maxsim batch|bit-identical|HIDEBUG" | tee -a "$OUT/summary.txt"
done
for combo in "0 0" "1 0"; do
  set -- $combo
  RAGLITE_FUSED_HI=$1 timeout 400 python scripts/bench_configs.py cfg5 > "$OUT/cfg5_hi$1.json" 2> "$OUT/cfg5_hi$1.err"
  echo "cfg5 fused_hi=$1 exit $?: $(tail -1 "$OUT/cfg5_hi$1.json" | cut -c1-420)" | tee -a "$OUT/summary.txt"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_cfg5" -o cfg5 -- python "$OLDPWD/scripts/bench_configs.py" cfg5 > "$OLDPWD/$OUT/prof_cfg5.json" 2> "$OLDPWD/$OUT/prof_cfg5.err" ); echo "prof cfg5 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof_cfg5" -name "*kernel_stats*" | head -1 | while read f; do head -14 "$f"; done
find "$OUT" -name "*kernel_trace*" -size +4M -delete
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.2f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms frac {r['roofline']['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    for k, v in (r.get('configs') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_query', 'ms_per_launch', 'ms', 'ms_per_batch', 'timing', 'error')})
    for k, v in (r.get('raglite_shaped') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'candidates_per_query', 'fallback_steps', 'check', 'error')})
except Exception as exc:
    print("  (no bench line)", exc)
PY
tail -5 "$OUT/bench.err"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
