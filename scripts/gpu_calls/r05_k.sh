#!/bin/bash
# Round 5, call k: cfg 5 against the sample stride of its first pass, now that the sample pass runs on the sixteen-group tile and the lists are cut by
# selection (round 4 measured 27 (the rule) 3.08, 40 3.03, 54 3.07, 80 3.15, 120 3.33 ms with the eight-group sample kernel and sorted lists).
set -u
OUT=gpurun_out/${1:-r05_k}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
for st in 0 40 54 64 80 0 54; do timeout 300 python scripts/bench_configs.py fused_topk_stride=$st cfg5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 stride=$st', d['ms_per_batch'], d['timing']['median_ms'], d['roofline'].get('kernel_ms'), d['candidates_per_query'])" | tee -a "$OUT/summary.txt"; done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
