#!/bin/bash
# One gpurun call for the switches built blind at the end of round 2 (no GPU minutes were left to measure them):
#   RAGLITE_HI_RNE=1          HI halves rounded to nearest (read when an index is created)
#   RAGLITE_HI_ONE_PRODUCT=0|1  approximate MaxSim pass with two / ONE fp16 MFMA products per multiply (read per call; 1 is the default)
#   RAGLITE_FUSED_HI=1        big-batch row top-k (cfg 5) over the HI image at one product per multiply (read per call)
# Stages: gated parity tests -> pass kernel time (rl_time_kernel kinds 5 / 6) -> headline bench under each combination.
# Usage (repo root on the GPU box): bash scripts/r3_experiments.sh [tag]
set -u
TAG=${1:-r03_exp}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
RAGLITE_TEST_EXPERIMENTAL=1 timeout 1200 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_hi_search.py tests/test_gpu_fused_topk.py -m gpu -q --timeout 600 > "$OUT/pytest_experimental.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -15 "$OUT/pytest_experimental.log"
timeout 600 python scripts/time_gemm_pass.py 1000000 20 3,5,6 > "$OUT/pass_times.txt" 2>&1; echo "pass times exit $?" | tee -a "$OUT/summary.txt"; cat "$OUT/pass_times.txt"
for combo in "0 0" "1 0" "0 1" "1 1"; do
  set -- $combo
  RAGLITE_HI_RNE=$1 RAGLITE_HI_ONE_PRODUCT=$2 RAGLITE_HI_DEBUG=${HI_DEBUG:-} timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-f16 \
      > "$OUT/bench_rne$1_one$2.json" 2> "$OUT/bench_rne$1_one$2.err"
  echo "bench rne=$1 one=$2 exit $?" | tee -a "$OUT/summary.txt"
  python - "$OUT/bench_rne$1_one$2.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  {r['value']:.0f} queries/s  {r['ms_per_step']:.2f} ms/step  pass {r['roofline']['kernel_ms']:.4f} ms  frac {r['roofline']['frac']:.3f}  recall {r.get('recall_at_100')}")
except Exception as exc:  # noqa: BLE001
    print("  (no bench line)", exc)
PY
done
for combo in "0 0" "1 0" "1 1"; do   # cfg 5: shipped fused top-k / over the HI image, one product / two products
  set -- $combo
  RAGLITE_FUSED_HI=$1 RAGLITE_FUSED_HI_TWO_PRODUCTS=$2 timeout 400 python scripts/bench_configs.py cfg5 > "$OUT/cfg5_hi$1_two$2.json" 2> "$OUT/cfg5_hi$1_two$2.err"
  echo "cfg5 fused_hi=$1 two_products=$2 exit $?: $(tail -1 "$OUT/cfg5_hi$1_two$2.json" | cut -c1-300)" | tee -a "$OUT/summary.txt"
done
# the deep-stream build of the one-product pass (maxsim_gemm_kernel HO: query fragments three slabs ahead, six-slot ring): parity, then time
RAGLITE_GEMM_DEEP=1 timeout 600 python -m pytest tests/test_gpu_hi_maxsim.py tests/test_gpu_fullsize.py -m gpu -q -x -k "hi_maxsim or fullsize_maxsim" --timeout 600 > "$OUT/pytest_deep.log" 2>&1
echo "pytest deep exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_deep.log"
RAGLITE_GEMM_DEEP=1 RAGLITE_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_fused_topk.py -m gpu -q -x -k fused_hi --timeout 600 > "$OUT/pytest_deep_fused.log" 2>&1
echo "pytest deep fused_hi exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_deep_fused.log"
gcc -O2 -std=c11 -Iinclude scripts/micro/r3_probe.c -o /tmp/r3_probe -Lraglite_amd/_lib -lraglite_hip -lm -Wl,-rpath,$PWD/raglite_amd/_lib
for deep in 0 1; do
  echo "== RAGLITE_GEMM_DEEP=$deep" | tee -a "$OUT/summary.txt"
  RAGLITE_GEMM_DEEP=$deep R3_PROBE_DEBUG=1 timeout 120 /tmp/r3_probe 2>&1 | tee "$OUT/probe_deep$deep.txt" | grep -E "pass kind|maxsim batch|bit-identical" | tee -a "$OUT/summary.txt"
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
