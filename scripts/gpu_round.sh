#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> bench -> rocprofv3 kernel trace (+ optional PMC pass).
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag] [stages]
#   stages: any of "smoke test bench prof pmc" (default: all but pmc)
set -u
TAG=${1:-r02}
STAGES=${2:-"smoke test bench prof"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start; stages: $STAGES" | tee "$OUT/summary.txt"
rocm-smi --showproductname 2>/dev/null | head -12 >> "$OUT/summary.txt"
nproc >> "$OUT/summary.txt"
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has smoke; then
  timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"
  tail -3 "$OUT/smoke.log"
fi
if has test; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
  tail -25 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 900 python bench.py --steps 40 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
  cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
fi
if has prof; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
  find "$OUT/prof" -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -15 "$f"; done
  # keep only the small summaries (the raw trace can be large)
  find "$OUT/prof" -name "*kernel_trace*" -size +8M -delete
fi
if has pmc; then
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/$OUT/pmc_fetch" -o bench -- python "$OLDPWD/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/pmc_fetch.err" ); echo "pmc fetch exit $?" | tee -a "$OUT/summary.txt"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OLDPWD/$OUT/pmc_mfma" -o bench -- python "$OLDPWD/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/pmc_mfma.err" ); echo "pmc mfma exit $?" | tee -a "$OUT/summary.txt"
  python scripts/summarize_pmc.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt"
  find "$OUT" -name "*.csv" -size +8M -delete
fi
echo "== $(date) done" | tee -a "$OUT/summary.txt"
