#!/bin/bash
# Round 3, GPU call G: the whole GPU suite with the promoted defaults (0 skips expected), smoke, the full bench line.
set -u
TAG=${1:-r03_g}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -15 "$OUT/pytest_gpu.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.2f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms frac {r['roofline']['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    print("  exact_fp32", {k: r['exact_fp32'].get(k) for k in ('value', 'ms_per_step')}, " f16_stored", {k: r['f16_stored'].get(k) for k in ('value', 'ms_per_step')})
    for k, v in (r.get('configs') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_query', 'ms_per_launch', 'ms', 'ms_per_batch', 'timing', 'error')})
    for k, v in (r.get('raglite_shaped') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'candidates_per_query', 'fallback_steps', 'error')})
except Exception as exc:
    print("  (no bench line)", exc)
PY
tail -3 "$OUT/bench.err"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
