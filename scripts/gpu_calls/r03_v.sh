#!/bin/bash
# Round 3, GPU call V: which part of maxsim_pp_kernel's main loop does not overlap -- skeletons WITHOUT the epilogue (DBG 128 keeps the MFMAs):
# no LDS fragment reads (136), no corpus DMAs (144), no query DMAs (160), no DMAs (176), MFMAs alone (184).
set -u
TAG=${1:-r03_v}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass,', round(r['ms_per_8_queries'],4), 'per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp 7 A=1
run pp_dbg128_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_dbg136_no_epilogue_no_reads 7 RAGLITE_PP_DBG=136
run pp_dbg144_no_epilogue_no_corpus_dma 7 RAGLITE_PP_DBG=144
run pp_dbg160_no_epilogue_no_query_dma 7 RAGLITE_PP_DBG=160
run pp_dbg176_no_epilogue_no_dma 7 RAGLITE_PP_DBG=176
run pp_dbg184_mfma_alone 7 RAGLITE_PP_DBG=184
run pp_dbg2_no_mfma 7 RAGLITE_PP_DBG=2
run pp_dbg64_corpus_from_l2 7 RAGLITE_PP_DBG=64
run pp_dbg192_no_epilogue_corpus_from_l2 7 RAGLITE_PP_DBG=192
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 300 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_pp.log"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
