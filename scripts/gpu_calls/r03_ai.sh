#!/bin/bash
# Round 3, GPU call AI: PMC counters of maxsim_pairs_kernel inside the headline step (own runs per counter set).
set -u
TAG=${1:-r03_ai}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$name" -o p -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/pmc_$name.err" ); echo "pmc $name exit $?" | tee -a "$OUT/summary.txt"
done
python scripts/summarize_pmc.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
grep "maxsim_pairs_kernel\|collect_above" "$OUT/pmc_summary.txt" | sed 's/(float const.*)//' | cut -c1-140
find "$OUT" -name "*.csv" -size +2M -delete
