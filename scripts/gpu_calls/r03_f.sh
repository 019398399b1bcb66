#!/bin/bash
# Round 3, GPU call F: does the deep-stream (HO) build of maxsim_gemm_kernel pay in the fused top-k over the HI image (cfg 5)?
set -u
TAG=${1:-r03_f}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
for deep in 0 1; do
  RAGLITE_FUSED_HI=1 RAGLITE_GEMM_DEEP=$deep timeout 400 python scripts/bench_configs.py cfg5 > "$OUT/cfg5_deep$deep.json" 2> "$OUT/cfg5_deep$deep.err"
  echo "cfg5 fused_hi deep=$deep exit $?: $(python -c "import json; r=json.loads(open('$OUT/cfg5_deep$deep.json').read().strip().splitlines()[-1]); print(r['ms_per_batch'], r['timing'], r['check'])")" | tee -a "$OUT/summary.txt"
done
RAGLITE_GEMM_DEEP=1 RAGLITE_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_fused_topk.py -m gpu -q -x --timeout 600 > "$OUT/pytest_deep_fused.log" 2>&1
echo "pytest deep fused exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_deep_fused.log"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
