#!/bin/bash
# Round 3, GPU call J: maxsim_pp2_kernel (two row streams, alternating phases): parity, pass time, skeletons.
set -u
TAG=${1:-r03_j}
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
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind$kind']; print(round(r['ms_per_8_queries'],4), 'ms per 8 queries')")" | tee -a "$OUT/summary.txt"
}
run pp_feed1 7 A=1
run pp2 8 A=1
run pp2_dbg1_no_scans 8 RAGLITE_PP_DBG=1
run pp2_dbg2_no_mfma 8 RAGLITE_PP_DBG=2
run pp2_dbg11_dma_only 8 RAGLITE_PP_DBG=11
run pp2_dbg48_no_dma 8 RAGLITE_PP_DBG=48
run pp2_dbg59_empty 8 RAGLITE_PP_DBG=59
for k in 1 2; do
  RAGLITE_PP_KERNEL=$k timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench_ppk$k.json" 2> "$OUT/bench_ppk$k.err"
  echo "bench PP_KERNEL=$k: $(python -c "
import json; r=json.loads(open('$OUT/bench_ppk$k.json').read().strip().splitlines()[-1]); print(round(r['value']), 'q/s', round(r['ms_per_step'],3), 'ms/step')")" | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
