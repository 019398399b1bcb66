#!/bin/bash
# Round 3, GPU call D: one feeder wave per SIMD (RAGLITE_PP_FEED=1); the empty-loop skeleton; what the L2 -> CU path delivers by instruction kind.
set -u
TAG=${1:-r03_d}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_PP_FEED=1 timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -q -x --timeout 600 > "$OUT/pytest_pp_feed1.log" 2>&1
echo "pytest pp feed 1 exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_pp_feed1.log"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/pass_$name.json" 2> /dev/null
  echo "$name: $(python -c "import json,sys; r=json.load(open('$OUT/pass_$name.json'))['kind7']; print(round(r['ms_per_pass'],4), 'ms per 16-query pass')")" | tee -a "$OUT/summary.txt"
}
run feed0 A=1
run feed0_dbg59_empty_loop RAGLITE_PP_DBG=59
run feed1 RAGLITE_PP_FEED=1
run feed1_dbg2_no_mfma RAGLITE_PP_FEED=1 RAGLITE_PP_DBG=2
run feed1_dbg11_dma_only RAGLITE_PP_FEED=1 RAGLITE_PP_DBG=11
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-inline-asm scripts/micro/l2_dma_rate.hip -o /tmp/l2_dma_rate 2>/dev/null && timeout 300 /tmp/l2_dma_rate 2>&1 | tee "$OUT/l2_dma_rate.txt" | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
