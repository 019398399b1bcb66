#!/bin/bash
# What one rank of an N-way sharded run does per step, measured on ONE GPU: the headline step over 1/N of the corpus (bench.py --rows).
# The whole-job rate of N ranks is at most QUERIES_PER_STEP / (this step's time + the exchange): the scaling the driver's N = 2, 4, 8 runs can show.
OUT=${1:-gpurun_out/shard_overhead}; mkdir -p "$OUT"
for rows in 1000000 500000 250000 125000; do
  timeout 300 python bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-f16 > "$OUT/rows_$rows.json" 2> "$OUT/rows_$rows.err"
  python - "$OUT/rows_$rows.json" $rows <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = int(sys.argv[2]); n = 1000000 // rows
print(f"rows {rows:8d} (1/{n}): {r['ms_per_step']:.3f} ms per 128-query step, pass {r['roofline']['kernel_ms']:.4f} ms, {r['value']:.0f} queries/s on one GPU -> "
      f"{n} ranks: <= {128 / r['ms_per_step'] * 1e3:.0f} queries/s = {128 / r['ms_per_step'] * 1e3 / n:.0f} per GPU; candidates {r.get('candidates_per_query')}")
PY
done
