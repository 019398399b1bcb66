#!/bin/bash
# Round 3, GPU call AD: the memory-budget test and the pooling tests after the removal of the cooperative kernel; kernel-by-kernel
# timeline of the single-query search (cfg 2): durations and launch gaps.
set -u
TAG=${1:-r03_ad}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_memory_budget.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "memory or pool or embed" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest.log"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o cfg2 -- python "$OLDPWD/scripts/cfg2_loop.py" > /dev/null 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
f=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1)
python scripts/kernel_gaps.py "$f" maxsim_stream | tee "$OUT/cfg2_kernel_gaps.txt"
rm -rf "$OUT/prof"
timeout 300 python scripts/bench_configs.py cfg2 > "$OUT/cfg2.json" 2> "$OUT/cfg2.err"; head -c 400 "$OUT/cfg2.json"; echo
echo "== $(date) done" | tee -a "$OUT/summary.txt"
