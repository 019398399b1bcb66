#!/bin/bash
# Round 4, GPU call AB (final): full verification on one fresh box -- smoke, the whole GPU suite, the full bench line (every block), rocprofv3
# kernel stats of the headline step, of cfg 5 and of cfg 2, PMC passes (own runs, kernel trace only) over the pass kernel and the cfg 5 candidate pass.
set -u
TAG=${1:-r04_ab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 >> "$OUT/summary.txt"; echo "host cpus: $(nproc)" >> "$OUT/summary.txt"
timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
tail -4 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = r['roofline']
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  launch {rf['kernel_ms']:.4f} ms ({rf.get('passes_per_launch')} passes: {rf.get('kernel_ms_per_pass', float('nan')):.4f} ms per pass) frac {rf['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    print("  memory", {k: v for k, v in (r.get('index_memory') or {}).items() if k != 'note'})
    print("  exact_fp32", {k: r['exact_fp32'].get(k) for k in ('value', 'ms_per_step')}, " f16_stored", {k: r['f16_stored'].get(k) for k in ('value', 'ms_per_step', 'candidates_per_query')})
    for k, v in (r.get('configs') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_query', 'ms', 'ms_per_batch', 'error')}, (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('kernel_frac'))
    for k, v in (r.get('raglite_shaped') or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ('value', 'ms_per_step', 'candidates_per_query', 'fallback_steps', 'error')})
    print("  tol", r.get('score_tolerance'))
    print("  cpu", r.get('cpu_baseline', {}).get('value'), r.get('cpu_baseline', {}).get('cores'))
except Exception as exc:
    print("  (no bench line)", exc)
PY
tail -2 "$OUT/bench.err"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" ); echo "prof exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/bench_kernel_stats.csv"; head -14 "$f" | cut -c1-170; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof5" -o cfg5 -- python "$OLDPWD/scripts/bench_configs.py" cfg5 > "$OLDPWD/$OUT/prof_cfg5.json" 2> "$OLDPWD/$OUT/prof5.err" ); echo "prof cfg5 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof5" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg5_kernel_stats.csv"; head -8 "$f" | cut -c1-170; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof2" -o cfg2 -- python "$OLDPWD/scripts/bench_configs.py" cfg2 > "$OLDPWD/$OUT/prof_cfg2.json" 2> "$OLDPWD/$OUT/prof2.err" ); echo "prof cfg2 exit $?" | tee -a "$OUT/summary.txt"
find "$OUT/prof2" -name "*kernel_stats*" | head -1 | while read f; do cp "$f" "$OUT/cfg2_kernel_stats.csv"; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-f16 --opt keep_image=0 --opt keep_hi_plane=0 > "$OUT/bench_slim.json" 2> "$OUT/bench_slim.err"; echo "bench slim exit $?" | tee -a "$OUT/summary.txt"
python - "$OUT/bench_slim.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"  slim: {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')} memory", {k: v for k, v in (r.get('index_memory') or {}).items() if k != 'note'})
except Exception as exc:
    print("  (no slim bench line)", exc)
PY
timeout 600 python scripts/shard_staged.py 1 8 2>/dev/null | tee "$OUT/shard_staged.txt"
find "$OUT" -name "*kernel_trace*" -delete
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "FETCH_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$name" -o p -- python "$OLDPWD/scripts/time_gemm_pass.py" 1000000 4 7 8 128 > /dev/null 2> "$OLDPWD/$OUT/pmc_$name.err" ); echo "pmc $name exit $?" | tee -a "$OUT/summary.txt"
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc5_$name" -o p -- python "$OLDPWD/scripts/bench_configs.py" cfg5 > /dev/null 2> "$OLDPWD/$OUT/pmc5_$name.err" ); echo "pmc cfg5 $name exit $?" | tee -a "$OUT/summary.txt"
done
python scripts/summarize_pmc.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1; grep -i "maxsim_pp\|counter\|===" "$OUT/pmc_summary.txt" | head -40
find "$OUT" -name "*.csv" -size +2M -delete
echo "== $(date) done" | tee -a "$OUT/summary.txt"
