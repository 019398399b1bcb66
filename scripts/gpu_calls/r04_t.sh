#!/bin/bash
# Round 4, GPU call T: the candidate pass of the fused row top-k (cfg 5) with QREG (query fragments to registers, records staged in LDS):
# its parity suite on the shipped library, then cfg 5 with either arrangement on the same box (experiment build: RAGLITE_PP_QREG=0 / 1).
set -u
OUT=gpurun_out/${1:-r04_t}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_fused_topk.py -m gpu -x -q > "$OUT/pytest_fused.log" 2>&1
echo "pytest fused_topk exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_fused.log" | tee -a "$OUT/summary.txt"
for q in 0 1 0 1; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PP_QREG=$q timeout 300 python scripts/bench_configs.py cfg5 2>/dev/null | tail -1 > "$OUT/cfg5_q$q.json"
  python - "$OUT/cfg5_q$q.json" $q <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
c = json.loads(open(sys.argv[1]).read())
print("  QREG=%s: cfg5 %.3f ms per batch, candidate pass %.3f ms, kernel frac %.3f, cand %s, check %s" % (sys.argv[2], c["ms_per_batch"], c["roofline"]["kernel_ms"], c["roofline"]["kernel_frac"], c["candidates_per_query"], c["check"]["recall_at_100"]))
PY
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
