#!/bin/bash
# Round 4, GPU call Z: FEED8 shipped (every wave fetches one corpus piece per slab): parity suites of both modes on the shipped library, then
# cfg 5 with FEED8 on / off (experiment build, RAGLITE_PP_ROWS_DBG=4096 = off) and the MaxSim pass likewise.
set -u
OUT=gpurun_out/${1:-r04_z}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_pp_pass.py tests/test_gpu_fused_topk.py tests/test_gpu_hi_maxsim.py -m gpu -x -q 2>&1 | tail -2 | tee -a "$OUT/summary.txt"
for d in 0 4096 0 4096; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_ROWS_DBG=$d timeout 300 python scripts/bench_configs.py cfg5 2>/dev/null | tail -1 > "$OUT/cfg5_$d.json"
  python - "$OUT/cfg5_$d.json" $d <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
c = json.loads(open(sys.argv[1]).read())
print("  ROWS_DBG=%s (4096 = FEED8 off): cfg5 %.3f ms per batch, candidate pass %.3f ms, kernel frac %.3f" % (sys.argv[2], c["ms_per_batch"], c["roofline"]["kernel_ms"], c["roofline"]["kernel_frac"]))
PY
done
for d in 0 4096; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_DBG=$d timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 128 2>/dev/null | tail -1 | sed "s/^/  MaxSim pass DBG=$d (4096 = FEED8 off): /" | cut -c1-170 | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
