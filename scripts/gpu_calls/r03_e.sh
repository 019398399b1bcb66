#!/bin/bash
# Round 3, GPU call E: slab timelines of maxsim_pp.hip (s_memtime stamps), MFMA-only variants.
set -u
TAG=${1:-r03_e}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind7']; print(round(r['ms_per_pass'],4), 'ms per 16-query pass')")" | tee -a "$OUT/summary.txt"
}
run feed0_dbg48_no_dma RAGLITE_PP_DBG=48
run feed0_dbg49_no_dma_no_scans RAGLITE_PP_DBG=49
run feed0_trace RAGLITE_PP_TRACE=1
grep PPTRACE "$OUT/pass_feed0_trace.err" > "$OUT/trace_feed0.txt"; head -42 "$OUT/trace_feed0.txt"
run feed1_trace RAGLITE_PP_TRACE=1 RAGLITE_PP_FEED=1
grep PPTRACE "$OUT/pass_feed1_trace.err" > "$OUT/trace_feed1.txt"; head -42 "$OUT/trace_feed1.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
