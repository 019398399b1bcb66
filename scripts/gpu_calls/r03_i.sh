#!/bin/bash
# Round 3, GPU call I: rocprofv3 kernel stats of the headline step (FEED 1 default), PMC passes of the shipped pass kernel.
set -u
TAG=${1:-r03_i}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python -c "
import json; r=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(r['value']), 'q/s', round(r['ms_per_step'],3), 'ms/step pass', round(r['roofline']['kernel_ms'],4))" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -25 "$f"; done
find "$OUT" -name "*kernel_trace*" -size +4M -delete
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "FETCH_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$name" -o p -- python "$OLDPWD/scripts/time_gemm_pass.py" 1000000 4 7 > /dev/null 2> "$OLDPWD/$OUT/pmc_$name.err" ); echo "pmc $name exit $?" | tee -a "$OUT/summary.txt"
done
python scripts/summarize_pmc.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1; head -60 "$OUT/pmc_summary.txt"
find "$OUT" -name "*.csv" -size +4M -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
