#!/bin/bash
# Last GPU seconds of round 2: the half-bytes MaxSim parity tests with the one-product pass as the default, then the rocprofv3
# kernel stats of a short bench run (the summary bench.py's live kernel time must agree with).
set -u
R=$PWD
OUT=$R/gpurun_out/r02_u
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 26 python -m pytest tests/test_gpu_hi_maxsim.py -m gpu -q -x -k "not experimental" > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"
tail -4 "$OUT/pytest.log"
( cd /tmp && timeout 28 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?"
f=$(find "$OUT/prof" -name "*kernel_stats*" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -8 "$f"; fi
find "$OUT/prof" -name "*kernel_trace*" -size +8M -delete 2>/dev/null
cut -c1-600 "$OUT/prof_bench.json"
