#!/bin/bash
# Round 3, GPU call AU: workgroups per query of maxsim_pairs_kernel (RAGLITE_PAIRS_PER_QUERY) -- is its time a tail of unequal workgroups?
set -u
OUT=gpurun_out/${1:-r03_au}; mkdir -p $OUT; export TMPDIR=/tmp
for pq in 0 1 2 8 16 32; do
  ( cd /tmp && RAGLITE_PAIRS_PER_QUERY=$pq timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_$pq" -o b -- python "$OLDPWD/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-f16 > "$OLDPWD/$OUT/bench_$pq.json" 2> /dev/null )
  f=$(find "$OUT/prof_$pq" -name "*kernel_stats*" | head -1)
  echo "per_query=$pq: $(grep maxsim_pairs_kernel "$f" | awk -F, '{print $(NF-5), "calls", $(NF-3)/1000, "us mean"}') ; $(python -c "import json; r=json.loads(open('$OUT/bench_$pq.json').read().strip().splitlines()[-1]); print(round(r['value']), 'q/s')")" | tee -a $OUT/summary.txt
  rm -rf "$OUT/prof_$pq"
done
