#!/bin/bash
# Round 4, GPU call N: the row-packing pairs kernel -- its bit-identity tests, the suites that re-score through it, then the headline step with
# either kernel on the same box (and the kernel stats of the packed run).
set -u
OUT=gpurun_out/${1:-r04_n}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_pairs_packed.py tests/test_gpu_hi_maxsim.py tests/test_gpu_pp_pass.py tests/test_gpu_sharded.py tests/test_gpu_shaped.py -m gpu -x -q > "$OUT/pytest_a.log" 2>&1
echo "pytest A exit $?" | tee -a "$OUT/summary.txt"; tail -5 "$OUT/pytest_a.log" | tee -a "$OUT/summary.txt"
show() {
  python - "$1" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("  %.0f q/s  %.3f ms/step  launch %.4f ms (%s passes) -> other %.3f ms  cand %s fb %s recall %s" % (r["value"], r["ms_per_step"], rf.get("kernel_ms", float("nan")), rf.get("passes_per_launch"), r["ms_per_step"] - rf.get("kernel_ms", 0.0), r.get("candidates_per_query"), r.get("fallback_steps"), r.get("recall_at_100")))
PY
}
for v in 1 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-f16 --opt pairs_packed=$v > "$OUT/bench_packed$v.json" 2> "$OUT/bench_packed$v.err"; echo "bench pairs_packed=$v exit $?" | tee -a "$OUT/summary.txt"; show "$OUT/bench_packed$v.json"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; grep -i "pairs\|maxsim_pp" "$f" | cut -c1-60,180-400 | tee -a "$OUT/summary.txt"; done
rm -rf "$OUT/prof"
timeout 600 python scripts/shard_staged.py 8 2>/dev/null | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
