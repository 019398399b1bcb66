#!/bin/bash
# Round 3, GPU call Y: every wave of maxsim_pp_kernel feeds (RAGLITE_PP_FEED=8: 4 query pieces -- its own fragments -- and 1 corpus piece per
# wave and slab) against one feeder per SIMD (default: 8 + 2 pieces on waves 0-3).
set -u
TAG=${1:-r03_y}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_PP_FEED=8 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 300 > "$OUT/pytest_pp_feed8.log" 2>&1
echo "pytest pp feed8 exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_pp_feed8.log"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass,', round(r['ms_per_8_queries'],4), 'per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp_feed4 7 A=1
run pp_feed8 7 RAGLITE_PP_FEED=8
run pp_feed4_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_feed8_no_epilogue 7 RAGLITE_PP_FEED=8 RAGLITE_PP_DBG=128
run pp_feed8_no_lag 7 RAGLITE_PP_FEED=8 RAGLITE_PP_DBG=1024
run pp_feed8_no_mfma 7 RAGLITE_PP_FEED=8 RAGLITE_PP_DBG=2
echo "== $(date) done" | tee -a "$OUT/summary.txt"
