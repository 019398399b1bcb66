#!/bin/bash
# Round 4, GPU call X: cfg 5 against the sample stride of its first pass (option fused_topk_stride; 0 = the built-in rule: 27 at k = 100).
set -u
OUT=gpurun_out/${1:-r04_x}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
for st in 0 40 54 80 120; do
  timeout 300 python scripts/bench_configs.py fused_topk_stride=$st cfg5 2>/dev/null | tail -1 > "$OUT/cfg5_stride$st.json"
  python - "$OUT/cfg5_stride$st.json" $st <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
c = json.loads(open(sys.argv[1]).read())
print("  stride %s: cfg5 %.3f ms per batch, candidate pass %.3f ms, cand %s, recall %s" % (sys.argv[2], c["ms_per_batch"], c["roofline"]["kernel_ms"], c["candidates_per_query"], c["check"]["recall_at_100"]))
PY
done
echo "== $(date) done" | tee -a "$OUT/summary.txt"
