#!/bin/bash
# Round 4, GPU call K: (1) all passes of a MaxSim batch in one launch (grid row = pass), (2) the shorter tail of the B <= 16 row search
# (threshold inside the collecting kernel, counters zeroed by the histogram launch, gather skips empty slots, transform inside the final
# merge), (3) 16-byte loads in the MaxSim threshold kernel, (4) the slim index bench.  Tests of everything touched, then the numbers.
set -u
OUT=gpurun_out/${1:-r04_k}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
EXP=$PWD/raglite_amd/_lib/libraglite_hip_exp.so
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 1200 python -m pytest tests/test_gpu_hi_search.py tests/test_gpu_hi_maxsim.py tests/test_gpu_pp_pass.py tests/test_gpu_sharded.py tests/test_gpu_gemm_pass.py tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest_a.log" 2>&1
echo "pytest A exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/pytest_a.log" | tee -a "$OUT/summary.txt"
for nq16 in 16 128; do
  timeout 300 python scripts/time_gemm_pass.py 1000000 20 7 8 $nq16 2>/dev/null | tail -1 | sed "s/^/  1M rows, $nq16 queries per launch: /" | tee -a "$OUT/summary.txt"
  timeout 300 python scripts/time_gemm_pass.py 125000 20 7 8 $nq16 2>/dev/null | tail -1 | sed "s/^/  125k rows, $nq16 queries per launch: /" | tee -a "$OUT/summary.txt"
done
show() {
  python - "$1" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rf = r["roofline"]
print("  %.0f q/s  %.3f ms/step  launch %.4f ms (%s passes) = %.4f ms/pass frac %.3f cand %s fb %s" % (r["value"], r["ms_per_step"], rf.get("kernel_ms", float("nan")), rf.get("passes_per_launch"), rf.get("kernel_ms_per_pass", float("nan")), rf["frac"], r.get("candidates_per_query"), r.get("fallback_steps")))
print("  memory", {k: v for k, v in (r.get("index_memory") or {}).items() if k != "note"})
PY
}
timeout 600 python bench.py --steps 10 --warmup 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default exit $?" | tee -a "$OUT/summary.txt"; show "$OUT/bench_default.json"
timeout 600 python bench.py --steps 10 --warmup 3 --opt keep_image=0 --opt keep_hi_plane=0 > "$OUT/bench_slim.json" 2> "$OUT/bench_slim.err"; echo "bench slim exit $?" | tee -a "$OUT/summary.txt"; show "$OUT/bench_slim.json"
for budget in 256 1024; do
  RAGLITE_HIP_LIB=$EXP RAGLITE_PAIRS_WG_BUDGET=$budget timeout 600 python bench.py --steps 10 --warmup 3 > "$OUT/bench_pairs$budget.json" 2> "$OUT/bench_pairs$budget.err"
  echo "bench pairs WG budget $budget exit $?" | tee -a "$OUT/summary.txt"; show "$OUT/bench_pairs$budget.json"
done
timeout 600 python scripts/bench_configs.py cfg2 cfg5 > "$OUT/cfg25.json" 2> "$OUT/cfg25.err"; echo "cfg2 cfg5 exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/cfg25.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"): continue
    r = json.loads(line)
    for name, c in (r.items() if "workload" not in r else [("", r)]):
        if isinstance(c, dict) and "workload" in c:
            print("  ", c["workload"], "value", c.get("value"), c.get("unit"), "ms", c.get("ms_per_query", c.get("ms_per_batch")), "kernel_ms", c["roofline"].get("kernel_ms"), "frac", c["roofline"].get("frac"))
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_cfg2" -o cfg2 -- python "$OLDPWD/scripts/bench_configs.py" cfg2 > /dev/null 2> "$OLDPWD/$OUT/prof_cfg2.err" ); echo "prof cfg2 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof_cfg2" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/cfg2_kernel_stats.csv"; rm -rf "$OUT/prof_cfg2"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
