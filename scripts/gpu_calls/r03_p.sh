#!/bin/bash
set -u
TAG=${1:-r03_p}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for d in 58 0; do
RAGLITE_PP_TRACE=1 RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 > "$OUT/t$d.json" 2> "$OUT/t$d.err"
echo "== DBG=$d"; grep PPTRACE "$OUT/t$d.err" | tee -a "$OUT/summary.txt"
done
