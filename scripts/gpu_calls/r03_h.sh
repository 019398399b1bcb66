#!/bin/bash
# Round 3, GPU call H: maxsim_pp.hip with the barrier half-way through the slab and the cheaper epilogue.
set -u
TAG=${1:-r03_h}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_pp.log"
RAGLITE_PP_FEED=1 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp_feed1.log" 2>&1
echo "pytest pp feed 1 exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_pp_feed1.log"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind7']; print(round(r['ms_per_pass'],4), 'ms per 16-query pass')")" | tee -a "$OUT/summary.txt"
}
run feed0 A=1
run feed1 RAGLITE_PP_FEED=1
run feed0_dbg1_no_scans RAGLITE_PP_DBG=1
run feed0_dbg2_no_mfma RAGLITE_PP_DBG=2
run feed0_dbg11_dma_only RAGLITE_PP_DBG=11
run feed0_dbg48_no_dma RAGLITE_PP_DBG=48
run feed0_dbg59_empty_loop RAGLITE_PP_DBG=59
run feed0_trace RAGLITE_PP_TRACE=1
grep PPTRACE "$OUT/pass_feed0_trace.err" > "$OUT/trace_feed0.txt"; head -26 "$OUT/trace_feed0.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
