#!/bin/bash
# Round 4, GPU call S: the shipped MaxSim pass is now QREG (query fragments to registers, results staged in LDS): whole GPU suite, the batch
# soak, the bench line, rocprofv3 stats of the headline step, shard 0 of an eight-way cut.
set -u
OUT=gpurun_out/${1:-r04_s}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/soak_hi_batch.py 120 51 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
timeout 900 python bench.py --steps 20 --warmup 5 --no-configs > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("  %.0f q/s  %.3f ms/step  launch %.4f ms (%s passes: %.4f per pass) frac %.3f cand %s fb %s recall %s" % (r["value"], r["ms_per_step"], rf["kernel_ms"], rf.get("passes_per_launch"), rf.get("kernel_ms_per_pass", float("nan")), rf["frac"], r.get("candidates_per_query"), r.get("fallback_steps"), r.get("recall_at_100")))
print("  f16_stored", {k: r.get("f16_stored", {}).get(k) for k in ("value", "ms_per_step", "kernel_ms")}, "tol", {k: (r.get("score_tolerance") or {}).get(k) for k in ("measured_rel_vs_f64", "abs_on_unit_norm")})
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; done; rm -rf "$OUT/prof"
timeout 600 python scripts/shard_staged.py 1 8 2>/dev/null | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
