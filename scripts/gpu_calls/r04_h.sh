#!/bin/bash
# Round 4, GPU call H: scripts/micro/simd_interference.hip -- what an instruction of one wave costs the MFMA stream of its SIMD partner.
set -u
OUT=gpurun_out/${1:-r04_h}
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/micro/simd_interference.hip -o /tmp/simd_interference 2>/dev/null
timeout 300 /tmp/simd_interference | tee "$OUT/simd_interference.txt"
