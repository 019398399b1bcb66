#!/bin/bash
# Round 3, GPU call X: waves 4-7 of maxsim_pp_kernel issue their fragment reads one MFMA group later than their SIMD partners (LAG); A/B against
# every wave running the same stream (DBG 1024), with and without the epilogue / the DMAs.
set -u
TAG=${1:-r03_x}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 300 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_pp.log"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass,', round(r['ms_per_8_queries'],4), 'per 8 queries')")" | tee -a "$OUT/summary.txt"
}
for extra in "$@"; do :; done
run pp 7 A=1
run pp_dbg1024_no_lag 7 RAGLITE_PP_DBG=1024
run pp_dbg128_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_dbg1152_no_lag_no_epilogue 7 RAGLITE_PP_DBG=1152
run pp_dbg176_no_epilogue_no_dma 7 RAGLITE_PP_DBG=176
run pp_dbg1200_no_lag_no_epilogue_no_dma 7 RAGLITE_PP_DBG=1200
run pp_dbg184_mfma_alone 7 RAGLITE_PP_DBG=184
echo "== $(date) done" | tee -a "$OUT/summary.txt"
