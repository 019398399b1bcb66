#!/bin/bash
# Round 4, GPU call O: what the packed pairs kernel's 0.16 ms are made of (experiment builds: WRONG results, timing only) -- rocprofv3 kernel
# stats of a short headline bench per variant: DBG 0 = shipped, 1 = no MFMAs, 2 = cache-resident rows, 3 = both, 4 = no query staging, 7 = all;
# and the split of a query's candidates over 1 / 2 / 4 / 8 workgroups.
set -u
OUT=gpurun_out/${1:-r04_o}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
run() {  # name, env...
  name=$1; shift
  ( cd /tmp && env RAGLITE_HIP_LIB=$EXP "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_$name" -o b -- python "$OLDPWD/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/$name.err" )
  f=$(find "$OUT/prof_$name" -name "*kernel_stats*" | head -1)
  echo "$name: $(grep -i 'pairs' "$f" | awk -F, '{print "calls", $(NF-6), "avg_us", $(NF-4)/1000, "min_us", $(NF-2)/1000, "max_us", $(NF-1)/1000}')" | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/prof_$name"
}
run dbg0 RAGLITE_PAIRS_DBG=0
run dbg1_no_mfma RAGLITE_PAIRS_DBG=1
run dbg2_cached_rows RAGLITE_PAIRS_DBG=2
run dbg3_neither RAGLITE_PAIRS_DBG=3
run dbg4_no_staging RAGLITE_PAIRS_DBG=4
run dbg7_nothing RAGLITE_PAIRS_DBG=7
run wg128 RAGLITE_PAIRS_WG_BUDGET=128
run wg512 RAGLITE_PAIRS_WG_BUDGET=512
run wg1024 RAGLITE_PAIRS_WG_BUDGET=1024
echo "== $(date) done" | tee -a "$OUT/summary.txt"
