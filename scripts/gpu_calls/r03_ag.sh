set -u
OUT=gpurun_out/r03_ag; mkdir -p $OUT; export TMPDIR=/tmp
run() { local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass')")" | tee -a "$OUT/summary.txt"; }
run pp 7 A=1
run pp_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_no_epilogue_no_query_dma 7 RAGLITE_PP_DBG=160
run pp_no_epilogue_no_query_reads 7 RAGLITE_PP_DBG=2176
run pp_no_epilogue_no_query_dma_no_query_reads 7 RAGLITE_PP_DBG=2208
