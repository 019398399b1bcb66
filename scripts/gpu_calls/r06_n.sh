#!/bin/bash
# Round 6, call n: verification of the round's code so far (smoke, whole GPU suite incl. the > 2^31-element module, full bench line, kernel stats)
# yardstick, rocprofv3 kernel stats of the headline step and of the vendor GEMM (kernel names), FETCH_SIZE of cfg 3 at both pool sizes.
set -u
TAG=${1:-r06_n}
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
timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python scripts/bench_summary.py "$OUT/bench.json" | tee -a "$OUT/summary.txt"
tail -2 "$OUT/bench.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -12 "$f" | cut -c1-170; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/profv" -o vendor -- python "$ROOT/scripts/vendor_gemm.py" > "$OUT/vendor_gemm.json" 2> "$OUT/profv.err" ); echo "prof vendor exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/profv" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/vendor_gemm_kernel_stats.csv"; head -8 "$f" | cut -c1-300 | tee -a "$OUT/summary.txt"; done
cat "$OUT/vendor_gemm.json" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_cfg3" -o p -- python "$ROOT/scripts/bench_configs.py" cfg3 > "$OUT/pmc_cfg3.json" 2> "$OUT/pmc_cfg3.err" ); echo "pmc cfg3 exit $?" | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_cfg3b" -o p -- python "$ROOT/scripts/bench_configs.py" cfg3_pool2g > "$OUT/pmc_cfg3b.json" 2> "$OUT/pmc_cfg3b.err" ); echo "pmc cfg3_pool2g exit $?" | tee -a "$OUT/summary.txt"
python scripts/summarize_pmc.py "$OUT/pmc_cfg3" 2>&1 | grep -E "^---|maxsim_cand" > "$OUT/pmc_cfg3_summary.txt"
python scripts/summarize_pmc.py "$OUT/pmc_cfg3b" 2>&1 | grep -E "^---|maxsim_cand" > "$OUT/pmc_cfg3_pool2g_summary.txt"
cat "$OUT/pmc_cfg3_summary.txt" "$OUT/pmc_cfg3_pool2g_summary.txt" | cut -c1-170 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
