#!/bin/bash
# Round 5, call d: tile_shapes.hip again, now with the current tile on v_mfma_f32_32x32x16_f16 (half as many MFMA instructions of twice the length).
set -u
OUT=gpurun_out/${1:-r05_d}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/tile_shapes.hip -o /tmp/tile_shapes 2> "$OUT/tile_shapes_build.err"; echo "build exit $?" | tee -a "$OUT/summary.txt"
timeout 300 /tmp/tile_shapes 2>&1 | tee -a "$OUT/summary.txt"
timeout 300 /tmp/tile_shapes 2>&1 | tee -a "$OUT/summary.txt"
echo "== $(date) done" | tee -a "$OUT/summary.txt"
