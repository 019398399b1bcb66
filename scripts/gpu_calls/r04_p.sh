#!/bin/bash
# Round 4, GPU call P: the pairs re-scoring alone at the headline shape (scripts/time_pairs.py: 128 queries x 100 / 78 candidates), shipped
# kernels and experiment builds (WRONG results, timing only): what the 0.16 ms are made of.
set -u
OUT=gpurun_out/${1:-r04_p}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
for nc in 100 78; do for p in 1 0; do timeout 200 python scripts/time_pairs.py $nc 20 $p 2>/dev/null | tail -1 | tee -a "$OUT/summary.txt"; done; done
for dbg in 1 2 3 4 7 8 16; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PAIRS_DBG=$dbg timeout 200 python scripts/time_pairs.py 100 20 1 2>/dev/null | tail -1 | sed "s/^/DBG $dbg: /" | tee -a "$OUT/summary.txt"
done
for b in 128 512 1024; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PAIRS_WG_BUDGET=$b timeout 200 python scripts/time_pairs.py 100 20 1 2>/dev/null | tail -1 | sed "s/^/WG budget $b: /" | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
