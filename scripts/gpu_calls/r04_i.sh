#!/bin/bash
# Round 4, GPU call I: shard 0 of an N-way cut on one GPU, per-step times with one candidate threshold for all shards (scripts/shard_staged.py)
# + the per-stage split and the one-GPU line for the projected efficiency.
set -u
OUT=gpurun_out/${1:-r04_i}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python scripts/shard_staged.py 1 2 4 8 2>/dev/null | tee "$OUT/shard_staged.txt"
timeout 600 python scripts/time_gemm_pass.py 125000 20 7 2>/dev/null | tail -1 | tee -a "$OUT/shard_staged.txt"
