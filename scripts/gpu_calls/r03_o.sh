#!/bin/bash
set -u
TAG=${1:-r03_o}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass')")" | tee -a "$OUT/summary.txt"
}
run dbg58_epilogue_and_loop 7 RAGLITE_PP_DBG=58
run dbg570_no_emit 7 RAGLITE_PP_DBG=570
run dbg2106_no_sum32 7 RAGLITE_PP_DBG=2106
run dbg4154_sum_without_swap 7 RAGLITE_PP_DBG=4154
run dbg314_no_store 7 RAGLITE_PP_DBG=314
echo "== $(date) done" | tee -a "$OUT/summary.txt"
