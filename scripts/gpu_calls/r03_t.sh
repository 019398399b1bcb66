#!/bin/bash
# Round 3, GPU call T: maxsim_pp.hip with the tile epilogue in registers (permlane transposes, running v_max under EXEC, DPP reductions); DBG 128 now keeps the MFMAs.
set -u
TAG=${1:-r03_t}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 300 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -12 "$OUT/pytest_pp.log"
run() { # name, kind, env...
  local name=$1; local kind=$2; shift; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 $kind > "$OUT/pass_$name.json" 2> "$OUT/pass_$name.err"
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_pass'],4), 'ms per pass,', round(r['ms_per_8_queries'],4), 'per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp 7 A=1
run pp_dbg128_no_epilogue 7 RAGLITE_PP_DBG=128
run pp_dbg2_no_mfma 7 RAGLITE_PP_DBG=2
run pp_dbg48_no_dma 7 RAGLITE_PP_DBG=48
run pp_dbg58_epilogue_and_loop 7 RAGLITE_PP_DBG=58
run pp_dbg186_empty 7 RAGLITE_PP_DBG=186
run gemm_one_product 6 A=1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench: $(python -c "
import json; r=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(r['value']), 'q/s', round(r['ms_per_step'],3), 'ms/step pass', round(r['roofline']['kernel_ms'],4), 'frac', round(r['roofline']['frac'],3))")" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_shaped.py tests/test_gpu_fullsize.py tests/test_gpu_rank_cut.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "hi_maxsim or shaped or fullsize_maxsim or staged_cut or duckdb" > "$OUT/pytest_more.log" 2>&1
echo "pytest more exit $?" | tee -a "$OUT/summary.txt"; tail -8 "$OUT/pytest_more.log"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
