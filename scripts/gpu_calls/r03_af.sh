#!/bin/bash
# Round 3, GPU call AF: maxsim_pp_kernel with the two waves of a SIMD in strictly alternating compute / load segments (RAGLITE_PP_PING=1)
# against the shipped stream (lagged partner).
set -u
TAG=${1:-r03_af}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_PP_PING=1 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 300 > "$OUT/pytest_pp_ping.log" 2>&1
echo "pytest pp ping exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_pp_ping.log"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass,', round(r['ms_per_8_queries'],4), 'per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp 7 A=1
run pp_ping 7 RAGLITE_PP_PING=1
run pp_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_ping_no_epilogue 7 RAGLITE_PP_PING=1 RAGLITE_PP_DBG=128
run pp_ping_no_epilogue_no_dma 7 RAGLITE_PP_PING=1 RAGLITE_PP_DBG=176
run pp_ping_mfma_alone 7 RAGLITE_PP_PING=1 RAGLITE_PP_DBG=184
run pp_ping_no_mfma 7 RAGLITE_PP_PING=1 RAGLITE_PP_DBG=2
echo "== $(date) done" | tee -a "$OUT/summary.txt"
