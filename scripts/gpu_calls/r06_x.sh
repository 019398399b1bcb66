#!/bin/bash
# Round 6, call x: verification of the round's LAST code -- smoke, the whole GPU suite (incl. the > 2^31-element module), the full bench line with
# every block, rocprofv3 kernel stats + the step's timeline, the N > 1 bench path on one GPU over gloo (2 and 4 ranks).
set -u
TAG=${1:-r06_x}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
echo "host cpus: $(nproc)" >> "$OUT/summary.txt"
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_scale_2g.py -m gpu -q --timeout 800 -x > "$OUT/pytest_scale.log" 2>&1; echo "pytest scale exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|Error\|assert" "$OUT/pytest_scale.log" | tail -12 | tee -a "$OUT/summary.txt"
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_scale_2g.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | tail -8 | tee -a "$OUT/summary.txt"
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python scripts/bench_summary.py "$OUT/bench.json" | tee -a "$OUT/summary.txt"
tail -2 "$OUT/bench.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -14 "$f" | cut -c1-170 | tee -a "$OUT/summary.txt"; done
f=$(find "$OUT/prof" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/step_timeline.py "$f" query_planes_kernel maxsim_pp_kernel | tee "$OUT/headline_timeline.txt" | tee -a "$OUT/summary.txt"
for w in cfg5:query_rows_planes:maxsim_pp_kernel cfg2:maxsim_stream_kernel:transform_bmax; do
  name=${w%%:*}; rest=${w#*:}; mark=${rest%%:*}; need=${rest#*:}
  rm -rf /tmp/tr_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- python "$ROOT/scripts/bench_configs.py" $name > /dev/null 2>&1 )
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/step_timeline.py "$f" $mark $need | tee "$OUT/${name}_timeline.txt" | tee -a "$OUT/summary.txt"
done
rm -rf /tmp/tr_one; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_one -o t -- python "$ROOT/scripts/time_one_query.py" 100 > "$OUT/one_query.json" 2>/dev/null )
f=$(find /tmp/tr_one -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/step_timeline.py "$f" maxsim_stream_kernel transform_bmax | tee "$OUT/one_query_timeline.txt" | tee -a "$OUT/summary.txt"
# the wide-embedder routes: kernel split of a 128-query MaxSim step at dim 1536 / 3072, and the randomised soak against float64
for d in 1536 3072; do
  rm -rf /tmp/w$d; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w$d -o t -- python "$ROOT/scripts/dev/wide_probe.py" $d $((460800000/d)) 128 2>&1 | grep "ms per" | tee -a "$OUT/wide_probe.txt" )
  f=$(find /tmp/w$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -9 "$f" | cut -c1-200 | tee -a "$OUT/wide_probe.txt"
done
timeout 400 python scripts/soak_wide.py 240 7 2>&1 | tail -2 | tee "$OUT/soak_wide.txt" | tee -a "$OUT/summary.txt"
bash scripts/test_multirank_one_gpu.sh 2>&1 | cut -c1-300 | tee "$OUT/multirank_one_gpu.txt" | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
