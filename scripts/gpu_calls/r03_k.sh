#!/bin/bash
set -u
TAG=${1:-r03_k}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_8_queries'],4), 'ms per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp2 8 A=1
run pp2_dbg59_empty 8 RAGLITE_PP_DBG=59
run pp2_dbg187_empty_no_epilogue 8 RAGLITE_PP_DBG=187
run pp2_dbg128_no_epilogue 8 RAGLITE_PP_DBG=128
run pp2_dbg130_no_epilogue_no_mfma 8 RAGLITE_PP_DBG=130
echo "== $(date) done" | tee -a "$OUT/summary.txt"
