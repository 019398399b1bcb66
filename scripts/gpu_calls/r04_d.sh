#!/bin/bash
# Round 4, GPU call D: the block epilogues INSIDE the main loop (maxsim_pp.hip STAG), first run: parity of both modes, then same-box A/B
# against the round-3 arrangement (experiments library, RAGLITE_PP_STAG=0) for the MaxSim pass and the cfg 5 candidate pass.
set -u
TAG=${1:-r04_d}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_pp_pass.py tests/test_gpu_fused_topk.py tests/test_gpu_hi_maxsim.py -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -6 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
for stag in 1 0; do
  for dbg in 0 128; do
    RAGLITE_HIP_LIB=$EXP RAGLITE_PP_STAG=$stag RAGLITE_PP_DBG=$dbg timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('  maxsim pass STAG=$stag DBG=$dbg: %.4f ms' % r['kind7']['ms_per_pass'])" | tee -a "$OUT/summary.txt"
  done
done
for stag in 1 0; do
  for dbg in 0 128; do
    RAGLITE_HIP_LIB=$EXP RAGLITE_PP_STAG=$stag RAGLITE_PP_ROWS_DBG=$dbg timeout 300 python scripts/bench_configs.py cfg5 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('  cfg5 STAG=$stag DBG=$dbg: batch %.3f ms, candidate pass %s ms, recall %s' % (r['ms_per_batch'], r['roofline'].get('kernel_ms'), r['check']['recall_at_100']))" | tee -a "$OUT/summary.txt"
  done
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms frac {r['roofline']['frac']:.3f} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
except Exception as exc:
    print("  (no bench line)", exc)
PY
echo "== $(date) done" | tee -a "$OUT/summary.txt"
