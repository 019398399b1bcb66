#!/bin/bash
# One rocprofv3 --pmc pass (kernel trace only, as gpurun requires) over a command; per-kernel means of the counters.
# Usage (repo root on the GPU box): bash scripts/pmc_cmd.sh <name> "<COUNTER ...>" <command...>  -> gpurun_out/<name>_pmc.txt
set -u
NAME=$1; COUNTERS=$2; shift 2
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
D=/tmp/pmc_$NAME
rm -rf "$D"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d "$D" -o p -- "$@" > "$D.log" 2>&1 )
if find "$D" -name "*counter_collection*.csv" 2>/dev/null | grep -q .; then
  python "$ROOT/scripts/summarize_pmc.py" "$D" > "$ROOT/gpurun_out/${NAME}_pmc.txt" 2>&1
  grep -v "at::native\|rocclr\|synth\|row_range\|row_norms\|presplit" "$ROOT/gpurun_out/${NAME}_pmc.txt" | head -40
else
  echo "no counter csv"; tail -5 "$D.log"
fi
