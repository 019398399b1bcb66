#!/bin/bash
# Round 3, GPU call C: maxsim_pp.hip with the DMAs issued during the first MFMA groups; ring depths; DMA-only skeletons.
set -u
TAG=${1:-r03_c}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp.log" 2>&1
echo "pytest pp exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_pp.log"
RAGLITE_PP_RINGS=44 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp44.log" 2>&1
echo "pytest pp rings 4+4 exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_pp44.log"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/pass_$name.json" 2> /dev/null
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind7']; print(round(r['ms_per_pass'],4), 'ms per 16-query pass')")" | tee -a "$OUT/summary.txt"
}
run default A=1
run dbg1 RAGLITE_PP_DBG=1
run dbg2 RAGLITE_PP_DBG=2
run dbg11 RAGLITE_PP_DBG=11
run dbg27_query_dma_only RAGLITE_PP_DBG=27
run dbg43_corpus_dma_only RAGLITE_PP_DBG=43
run rings44 RAGLITE_PP_RINGS=44
run rings44_dbg11 RAGLITE_PP_RINGS=44 RAGLITE_PP_DBG=11
echo "== $(date) done" | tee -a "$OUT/summary.txt"
