#!/bin/bash
# FETCH_SIZE (counters + kernel trace only) of the wide-embedder kernels: a 128-query MaxSim step at dim 1536 / 3072, one cosine query over 650 k x 1536.
# Usage: bash scripts/pmc_wide.sh TAG ; results: gpurun_out/TAG/pmc_wide_summary.txt
set -u
TAG=${1:-pmc_wide}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for cmd in "scripts/dev/wide_probe.py 1536 300000 128" "scripts/dev/wide_probe.py 3072 150000 128" "scripts/dev/wide_rows_probe.py cosine"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_wide_$i" -o p -- python $ROOT/$cmd 2>&1 | grep "ms per" )
  echo "pmc wide $i ($cmd) exit $?"
done
python "$ROOT/scripts/summarize_pmc.py" "$OUT" 2>&1 | grep -E "^---|maxsim|scan_rows|pivot|transform" | cut -c1-160 > "$OUT/pmc_wide_summary.txt"
cat "$OUT/pmc_wide_summary.txt"
find "$OUT" -name "*.csv" -size +4M -delete
