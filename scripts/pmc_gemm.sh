#!/bin/bash
# PMC passes over scripts/time_gemm_pass.py (kind 3 = the eight-query pass kernel only).  Usage: bash scripts/pmc_gemm.sh TAG ["ENV=.."]
set -u
TAG=${1:-pmc_gemm}; ENVV=${2:-}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name, counters...
  local name=$1; shift
  ( cd /tmp && env $ENVV timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o t -- python "$ROOT/scripts/time_gemm_pass.py" 1000000 5 3 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" )
  echo "pmc $name exit $?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
python "$ROOT/scripts/summarize_pmc.py" "$OUT" 2>&1 | grep -E "^---|maxsim_gemm" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt"
find "$OUT" -name "*.csv" -size +2M -delete
