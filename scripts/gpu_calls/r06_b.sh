#!/bin/bash
# Round 6, call b: the few-queries MaxSim route over the HI plane (tests/test_gpu_hi_few.py), the two tests call a failed, the whole GPU suite,
# the bench line with its one_query block.
set -u
TAG=${1:-r06_b}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_hi_few.py "tests/test_gpu_scale_2g.py::test_append_across_the_2g_boundary_equals_an_index_built_at_once" "tests/test_gpu_memory_budget.py::test_prepare_builds_the_lazy_images_outside_the_hot_path_and_a_skipped_image_is_retried" -m gpu -q --timeout 800 > "$OUT/pytest_new.log" 2>&1; echo "pytest new exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR\|^E  " "$OUT/pytest_new.log" | tail -30 | cut -c1-300 | tee -a "$OUT/summary.txt"
timeout 1700 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_scale_2g.py --deselect tests/test_gpu_hi_few.py > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?" | tee -a "$OUT/summary.txt"
grep -a "passed\|failed\|^FAILED\|^ERROR" "$OUT/pytest_gpu.log" | tail -12 | tee -a "$OUT/summary.txt"
timeout 1200 python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench exit $?" | tee -a "$OUT/summary.txt"
python scripts/bench_summary.py "$OUT/bench.json" | tee -a "$OUT/summary.txt"
python - "$OUT/bench.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  one_query", json.dumps(r.get("one_query"))[:1500])
PY
tail -2 "$OUT/bench.err"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
