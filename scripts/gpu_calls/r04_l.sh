#!/bin/bash
# Round 4, GPU call L: the eight-launch B <= 16 row search (candidates listed by the selection's final kernel, guarded full pass riding on the
# re-scoring launch, one-launch guarded selection) -- the whole GPU suite, cfg 2 with its kernel stats, the slim-index bench.
set -u
OUT=gpurun_out/${1:-r04_l}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest -m gpu exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 600 python scripts/bench_configs.py cfg2 > "$OUT/cfg2.json" 2> "$OUT/cfg2.err"; echo "cfg2 exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/cfg2.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        c = json.loads(line)
        print("  ", c["workload"], "value", c.get("value"), c.get("unit"), "ms", c.get("ms_per_query"), "kernel_ms", c["roofline"].get("kernel_ms"), "frac", c["roofline"].get("frac"), c["check"])
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_cfg2" -o cfg2 -- python "$OLDPWD/scripts/bench_configs.py" cfg2 > /dev/null 2> "$OLDPWD/$OUT/prof_cfg2.err" ); echo "prof cfg2 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof_cfg2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/cfg2_kernel_stats.csv"; rm -rf "$OUT/prof_cfg2"
timeout 600 python bench.py --steps 10 --warmup 3 --opt keep_image=0 --opt keep_hi_plane=0 > "$OUT/bench_slim.json" 2> "$OUT/bench_slim.err"; echo "bench slim exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench_slim.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("  %.0f q/s  %.3f ms/step  launch %.4f ms (%s passes) = %.4f ms/pass frac %.3f cand %s fb %s recall %s" % (r["value"], r["ms_per_step"], rf.get("kernel_ms", float("nan")), rf.get("passes_per_launch"), rf.get("kernel_ms_per_pass", float("nan")), rf["frac"], r.get("candidates_per_query"), r.get("fallback_steps"), r.get("recall_at_100")))
print("  memory", {k: v for k, v in (r.get("index_memory") or {}).items() if k != "note"})
PY
echo "== $(date) done" | tee -a "$OUT/summary.txt"
