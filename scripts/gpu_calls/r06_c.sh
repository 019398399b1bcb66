#!/bin/bash
# Round 6, call c: the one-launch block top-k (tests, A/B on the headline step), the per-kernel split of the one-query route, the embedder's profile.
set -u
TAG=${1:-r06_c}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_topk_block.py -m gpu -q --timeout 600 > "$OUT/pytest_new.log" 2>&1; echo "pytest new exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_new.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
for o in 1 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm --no-cpu-baseline --no-f16 --opt topk_block=$o > "$OUT/bench_block$o.json" 2> "$OUT/bench_block$o.err"; echo "bench topk_block=$o exit $?" | tee -a "$OUT/summary.txt"
  python scripts/bench_summary.py "$OUT/bench_block$o.json" | head -2 | tee -a "$OUT/summary.txt"
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -o one -- python "$ROOT/scripts/time_one_query.py" 200 > "$OUT/one_query.json" 2> "$OUT/prof1.err" ); echo "prof one_query exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof1" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/one_query_kernel_stats.csv"; cut -c1-110,200-330 "$f" | head -28 | tee -a "$OUT/summary.txt"; done
cat "$OUT/one_query.json" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/time_one_query.py 200 | tee -a "$OUT/summary.txt"
timeout 300 python scripts/bench_configs.py cfg5 cfg2 > "$OUT/cfg52.json" 2> "$OUT/cfg52.err"; python - "$OUT/cfg52.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    print("  ", r["workload"][:40], {k: r.get(k) for k in ("value", "ms_per_batch", "ms_per_query")})
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/profe" -o embed -- python "$ROOT/scripts/bench_embed.py" 4000 > "$OUT/embed.json" 2> "$OUT/profe.err" ); echo "prof embed exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/profe" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/embed_kernel_stats.csv"; cut -c1-160 "$f" | head -16 | tee -a "$OUT/summary.txt"; done
cat "$OUT/embed.json" | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
