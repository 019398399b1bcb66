#!/bin/bash
# Round 5, call g: rocprofv3 kernel stats + PMC passes (own runs, kernel trace only) over the bench's headline block, the traffic record
# filed under the kernel's own name (profiles/traffic.json "by_kernel"), cfg 5 / cfg 2 kernel stats.
set -u
TAG=${1:-r05_g}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
BENCH="python $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- $BENCH > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -6 "$f" | cut -c1-160; done
run() { local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o bench -- $BENCH > /dev/null 2> "$OUT/pmc_$name.err" ); echo "pmc $name exit $?" | tee -a "$OUT/summary.txt"; }
run fetch FETCH_SIZE
run busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
python scripts/summarize_pmc.py "$OUT" --traffic-json "$OUT/traffic.json" --passes-per-launch 8 --source "profiles/${TAG}_pmc_summary.txt" 2>&1 | grep -E "^---|maxsim_pp|mfma_f16|maxsim_pairs|topk_|collect" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt" | cut -c1-200
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof5" -o cfg5 -- python '$ROOT'/scripts/bench_configs.py cfg5 > "$OUT/cfg5.json" 2> "$OUT/prof5.err" ); echo "prof cfg5 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof5" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg5_kernel_stats.csv"; done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2" -o cfg2 -- python '$ROOT'/scripts/bench_configs.py cfg2 > "$OUT/cfg2.json" 2> "$OUT/prof2.err" ); echo "prof cfg2 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof2" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg2_kernel_stats.csv"; done
find "$OUT" -name "*.csv" -size +4M -delete
find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
