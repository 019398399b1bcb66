#!/bin/bash
# Round 4, GPU call C: where the candidate pass of cfg 5 (maxsim_pp MODE 2) spends its time -- experiment builds (wrong results):
# no tile epilogue / no flush / neither, against the shipped kernel, same box; and the MaxSim pass (kind 7) for the same comparison.
set -u
TAG=${1:-r04_c}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
for dbg in 0 128 512 640; do
  RAGLITE_HIP_LIB=$PWD/raglite_amd/_lib/libraglite_hip_exp.so RAGLITE_PP_ROWS_DBG=$dbg timeout 300 python scripts/bench_configs.py cfg5 > "$OUT/cfg5_dbg$dbg.json" 2> "$OUT/cfg5_dbg$dbg.err"
  python - "$OUT/cfg5_dbg$dbg.json" $dbg <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  RAGLITE_PP_ROWS_DBG={sys.argv[2]}: batch {r['ms_per_batch']:.3f} ms, candidate pass {r['roofline'].get('kernel_ms')} ms, recall {r['check']['recall_at_100']}")
except Exception as exc:
    print("  (no line)", exc)
PY
done
for dbg in 0 128; do
  RAGLITE_HIP_LIB=$PWD/raglite_amd/_lib/libraglite_hip_exp.so RAGLITE_PP_DBG=$dbg timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 2>&1 | tail -2 | sed "s/^/  RAGLITE_PP_DBG=$dbg: /" | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
