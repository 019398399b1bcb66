#!/bin/bash
# Round 4, GPU call E: the candidate pass of cfg 5 in two rounds (thresholds tightened after 3/16 of the row tiles): parity, then cfg 5.
set -u
TAG=${1:-r04_e}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fused_topk.py tests/test_gpu_hi_search.py -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
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
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o cfg5 -- python "$OLDPWD/scripts/bench_configs.py" cfg5 > "$OLDPWD/$OUT/prof_cfg5.json" 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg5_kernel_stats.csv"; head -16 "$f" | cut -c1-200 | tee -a "$OUT/summary.txt"; done
find "$OUT/prof" -name "*kernel_trace*" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
