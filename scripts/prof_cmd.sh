#!/bin/bash
# rocprofv3 kernel-trace summary of one command, safe for a gpurun call (every step under `timeout`, no stdin reads).
# Usage (repo root on the GPU box): bash scripts/prof_cmd.sh <name> <command...>  -> gpurun_out/<name>_kernel_stats.csv (top 16 rows)
set -u
NAME=$1; shift
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
D=/tmp/prof_$NAME
rm -rf "$D"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o p -- "$@" > "$D.log" 2>&1 )
f=$(find "$D" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then
  head -17 "$f" | cut -c1-240 > "$ROOT/gpurun_out/${NAME}_kernel_stats.csv"
  cat "$ROOT/gpurun_out/${NAME}_kernel_stats.csv"
else
  echo "no kernel_stats.csv"; tail -5 "$D.log"
fi
