#!/bin/bash
# Round 3, GPU call AV: rocprofv3 kernel stats over the cfg 5 block (1000 queries x 1.25 M rows, exact top-100, the shipped fused top-k over
# the HI image) next to its own timing line: the kernels' sum against ms_per_batch.
set -u
OUT=gpurun_out/${1:-r03_av}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python scripts/bench_configs.py cfg5 > $OUT/cfg5.json 2> $OUT/cfg5.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o cfg5 -- python "$OLDPWD/scripts/bench_configs.py" cfg5 > "$OLDPWD/$OUT/cfg5_profiled.json" 2> /dev/null )
f=$(find "$OUT/prof" -name "*kernel_stats*" | head -1); cp "$f" $OUT/cfg5_kernel_stats.csv; rm -rf $OUT/prof
python - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
r = json.loads(open(f"{out}/cfg5.json").read().strip().splitlines()[-1])
p = json.loads(open(f"{out}/cfg5_profiled.json").read().strip().splitlines()[-1])
print("cfg5 ms_per_batch", r["ms_per_batch"], "timing", r["timing"], "| under rocprofv3:", p["ms_per_batch"])
rows = list(csv.DictReader(open(f"{out}/cfg5_kernel_stats.csv")))
iters = p["timing"]["iters"] + p["timing"]["warmup"]
tot = 0.0
for x in rows:
    calls, total = int(x["Calls"]), float(x["TotalDurationNs"])
    if calls >= iters and calls % iters == 0 or "maxsim_gemm_kernel" in x["Name"] or "row_dots" in x["Name"]:
        per = total / iters / 1e6
        tot += per
        if per > 0.005: print(f"  {x['Name'][:70]:70s} {calls:5d} calls  {per:7.3f} ms per batch")
print(f"  sum of the per-batch kernels: {tot:.3f} ms")
PY
