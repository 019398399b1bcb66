#!/bin/bash
# A/B of kernel variants selected through environment variables.  Usage: bash scripts/ab_bench.sh TAG "VAR=a VAR=b ..."
set -u
TAG=${1:-ab}
shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' =' '__')
  for rep in 1 2; do
    env $cfg timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_${name}_$rep.json" 2> "$OUT/bench_${name}_$rep.err"
    echo "$cfg rep$rep: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${name}_$rep.json')); print('q/s=%.1f kernel_ms=%.4f frac=%.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))" 2>&1 | tail -1)"
  done
done
