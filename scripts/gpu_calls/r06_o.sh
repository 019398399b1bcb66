#!/bin/bash
# Round 6, call o: the full bench line (call n's bench ran into the hashing tokenizer's full table at cfg 4's scale: fixed) + rocprofv3 kernel stats
# of the headline step, PMC passes over it.
set -u
TAG=${1:-r06_o}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
echo "host cpus: $(nproc)" >> "$OUT/summary.txt"
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python scripts/bench_summary.py "$OUT/bench.json" | tee -a "$OUT/summary.txt"
tail -2 "$OUT/bench.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -14 "$f" | cut -c1-170 | tee -a "$OUT/summary.txt"; done
f=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/step_timeline.py "$f" query_planes_kernel maxsim_pp_kernel | tee "$OUT/headline_timeline.txt" | tee -a "$OUT/summary.txt"
bash scripts/pmc.sh $TAG "" 2>&1 | tail -30 | cut -c1-200 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
