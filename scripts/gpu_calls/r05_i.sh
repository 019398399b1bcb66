#!/bin/bash
# Round 5, call i: the sample pass of the fused top-k on the sixteen-group tile (maxsim_pp.hip MODE 1): parity, then cfg 5 A/B on one box.
set -u
TAG=${1:-r05_i}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fused_topk.py tests/test_gpu_fullsize.py -q -x --timeout 600 > "$OUT/pytest_fused.log" 2>&1; echo "fused tests exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_fused.log" | tail -12 | tee -a "$OUT/summary.txt"
for v in 1 0 1 0; do timeout 300 python scripts/bench_configs.py list_select=$v cfg5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 list_select=$v', d['ms_per_batch'], d['timing']['median_ms'], d['roofline'].get('kernel_ms'), d['candidates_per_query'], d['check'])" | tee -a "$OUT/summary.txt"; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof5" -o cfg5 -- python "$ROOT/scripts/bench_configs.py" cfg5 > "$OUT/prof_cfg5.json" 2> "$OUT/prof5.err" ); echo "prof cfg5 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof5" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg5_kernel_stats.csv"; head -12 "$f" | cut -c1-150 | tee -a "$OUT/summary.txt"; done
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
