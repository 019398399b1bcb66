#!/bin/bash
# PMC passes over bench.py (kernel-trace + counters only, one counter group per run).
# Usage: bash scripts/pmc.sh TAG "ENV=..." ; results: gpurun_out/TAG/pmc_<group>/..., summary in pmc_summary.txt
set -u
TAG=${1:-pmc}; ENVV=${2:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name, counters...
  local name=$1; shift
  ( cd /tmp && env $ENVV timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o bench -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs > /dev/null 2> "$OUT/pmc_$name.err" )
  echo "pmc $name exit $?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
python "$OLDPWD/scripts/summarize_pmc.py" "$OUT" 2>&1 | grep -E "^---|maxsim|topk" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt"
find "$OUT" -name "*.csv" -size +4M -delete
