#!/bin/bash
# Round 5, call h (verification): smoke, the whole GPU suite, the full bench line (every block), rocprofv3 kernel stats of the headline step,
# of cfg 5 and of cfg 2, the bench with every image built at once (lazy_images = 0), shard 0 of an eight-way cut, one PMC pass over the pass kernel.
set -u
TAG=${1:-r05_h}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
echo "host cpus: $(nproc)" >> "$OUT/summary.txt"
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | tail -6 | tee -a "$OUT/summary.txt"
timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = r['roofline']
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  launch {rf['kernel_ms']:.4f} ms ({rf.get('passes_per_launch')} passes: {rf.get('kernel_ms_per_pass', float('nan')):.4f} ms per pass) frac {rf['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    print("  sustained", {k: v for k, v in (rf.get('sustained') or {}).items() if k not in ('how',)})
    print("  traffic", rf.get('traffic'), "|", (rf.get('traffic_source') or '')[:120])
    print("  memory", {k: v for k, v in (r.get('index_memory') or {}).items() if k != 'note'})
    print("  exact_fp32", {k: r['exact_fp32'].get(k) for k in ('value', 'ms_per_step', 'frac')}, " f16_stored", {k: r['f16_stored'].get(k) for k in ('value', 'ms_per_step', 'candidates_per_query')})
    print("  f16_queries", {k: r['f16_queries'].get(k) for k in ('value', 'ms_per_step', 'route', 'fallback', 'score_max_rel_err', 'recall_at_100_slab')})
    for k, v in (r.get('configs') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_query', 'ms', 'ms_per_batch', 'error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('kernel_frac'))
    for k, v in (r.get('raglite_shaped') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'candidates_per_query', 'fallback_steps', 'error')}, "f16 queries:", {kk: (v.get('f16_queries') or {}).get(kk) for kk in ('value', 'route')})
    print("  tol", r.get('score_tolerance'))
    print("  cpu", r.get('cpu_baseline', {}).get('value'), r.get('cpu_baseline', {}).get('cores'), "| fraction_check:", r.get('fraction_check'))
except Exception as exc:
    print("  (no bench line)", exc)
PY
tail -2 "$OUT/bench.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OUT/prof_bench.json" 2> "$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -8 "$f" | cut -c1-170; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof5" -o cfg5 -- python "$ROOT/scripts/bench_configs.py" cfg5 > "$OUT/prof_cfg5.json" 2> "$OUT/prof5.err" ); echo "prof cfg5 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof5" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg5_kernel_stats.csv"; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2" -o cfg2 -- python "$ROOT/scripts/bench_configs.py" cfg2 > "$OUT/prof_cfg2.json" 2> "$OUT/prof2.err" ); echo "prof cfg2 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof2" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg2_kernel_stats.csv"; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-f16 --opt lazy_images=0 > "$OUT/bench_eager.json" 2> "$OUT/bench_eager.err"; echo "bench eager exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench_eager.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  lazy_images=0: {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  cand {r.get('candidates_per_query')} memory", {k: v for k, v in (r.get('index_memory') or {}).items() if k != 'note'})
except Exception as exc:
    print("  (no eager bench line)", exc)
PY
timeout 600 python scripts/shard_staged.py 1 8 2>/dev/null | tee "$OUT/shard_staged.txt" | tail -4 | tee -a "$OUT/summary.txt"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_busy" -o p -- python "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > /dev/null 2> "$OUT/pmc_busy.err" ); echo "pmc busy exit $?" | tee -a "$OUT/summary.txt"
python scripts/summarize_pmc.py "$OUT" 2>&1 | grep -E "^---|maxsim_pp|mfma_f16" > "$OUT/pmc_summary.txt"; cat "$OUT/pmc_summary.txt" | cut -c1-170 | tee -a "$OUT/summary.txt"
find "$OUT" -name "*kernel_trace*" -delete; find "$OUT" -name "*.csv" -size +2M -delete; find "$OUT" -name "*.db" -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
