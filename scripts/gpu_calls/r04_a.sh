#!/bin/bash
# Round 4, GPU call A: first run of (1) the candidate pass of cfg 5 on the sixteen-group tile (maxsim_pp MODE 2), (2) the exact-k-th
# candidate threshold of MaxSim batches, (3) the route options replacing every getenv, (4) the adversarial bound tests.
# Parity of everything touched, then same-box A/B of cfg 5 (fused_pp 1 / 0) and of the headline (exact_kth_threshold 1 / 0).
set -u
TAG=${1:-r04_a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fused_topk.py tests/test_gpu_pp_pass.py tests/test_gpu_hi_maxsim.py -m gpu -q -x --timeout 600 -s > "$OUT/pytest_a.log" 2>&1
echo "pytest A exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_a.log"; grep -h "^\[" "$OUT/pytest_a.log" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_hi_search.py tests/test_gpu_memory_budget.py tests/test_abi.py tests/test_gpu_sharded.py -m gpu -q --timeout 600 -s > "$OUT/pytest_b.log" 2>&1
echo "pytest B exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_b.log"; grep -h "^\[" "$OUT/pytest_b.log" | sort | uniq | head -40 | tee -a "$OUT/summary.txt"
for opt in "fused_pp=1" "fused_pp=0"; do
  timeout 300 python scripts/bench_configs.py $opt cfg5 > "$OUT/cfg5_$opt.json" 2> "$OUT/cfg5_$opt.err"; echo "cfg5 $opt exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT/cfg5_$opt.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ", {k: r.get(k) for k in ("ms_per_batch", "timing", "candidates_per_query", "check")})
    print("  ", r.get("roofline"))
except Exception as exc:
    print("  (no line)", exc)
PY
  tail -2 "$OUT/cfg5_$opt.err"
done
for opt in "exact_kth_threshold=1" "exact_kth_threshold=0"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 --split --opt $opt > "$OUT/bench_$opt.json" 2> "$OUT/bench_$opt.err"; echo "bench $opt exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT/bench_$opt.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms frac {r['roofline']['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    print("  split", r.get("rank_split", {}).get("ranks"))
    print("  tol", r.get("score_tolerance"))
except Exception as exc:
    print("  (no bench line)", exc)
PY
  tail -2 "$OUT/bench_$opt.err"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
