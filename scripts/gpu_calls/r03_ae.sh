#!/bin/bash
# Round 3, GPU call AE (also used for AJ: the software-pipelined maxsim_pairs_kernel): tests of every bound-filtered path and the rerank paths, kernel stats of the headline step, bench.
# the tests of every bound-filtered path, kernel stats of the headline step, bench.
set -u
TAG=${1:-r03_ae}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "hi_ or hi_maxsim or half_bytes or shaped or pp_ or fused or memory or fullsize or rerank or pairs or generic or maxsim" > "$OUT/pytest.log" 2>&1
echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest.log"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -14 "$f" | cut -c1-150; done
rm -rf "$OUT/prof"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench: $(python -c "
import json; r=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(round(r['value']), 'q/s', round(r['ms_per_step'],3), 'ms/step pass', round(r['roofline']['kernel_ms'],4), 'frac', round(r['roofline']['frac'],3), 'cand', r.get('candidates_per_query'), 'fb', r.get('fallback_steps'))")" | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
