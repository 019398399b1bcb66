#!/bin/bash
# Round 4, GPU call F: the candidate pass of cfg 5 piece by piece on ONE box: one round / two rounds, and the timing skeletons of the
# kernel (experiments library): no block epilogues, no flush.
set -u
TAG=${1:-r04_f}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fused_topk.py -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
line() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $1: batch %.3f ms, candidate pass %.3f ms, candidates %.0f (max %s), recall %s' % (r['ms_per_batch'], r['roofline'].get('kernel_ms') or 0, r['candidates_per_query']['mean'] or 0, r['candidates_per_query']['max'], r['check']['recall_at_100']))" | tee -a "$OUT/summary.txt"; }
for rounds in 1 0; do
  for dbg in 0 128 512 640; do
    RAGLITE_HIP_LIB=$EXP RAGLITE_PP_ROWS_DBG=$dbg timeout 300 python scripts/bench_configs.py fused_two_rounds=$rounds cfg5 2>/dev/null | line "two_rounds=$rounds DBG=$dbg"
  done
done
timeout 300 python scripts/bench_configs.py fused_pp=0 cfg5 2>/dev/null | line "eight-group tile"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
