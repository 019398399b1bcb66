#!/bin/bash
set -u
OUT=gpurun_out/${1:-r04_u}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_pp_pass.py -m gpu -x -q -k "staged or one_launch" 2>&1 | tail -5 | tee "$OUT/summary.txt"
