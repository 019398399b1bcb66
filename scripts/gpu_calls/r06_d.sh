#!/bin/bash
# Round 6, call d: launch-by-launch timelines (durations + idle gaps) of the headline step, cfg 5, cfg 2 and the one-query MaxSim route;
# probe of the flash-attention backends torch offers for the embedder's shape.
set -u
TAG=${1:-r06_d}
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) start" | tee "$OUT/summary.txt"
trace() {  # name, mark, need, command...
  local name=$1 mark=$2 need=$3; shift 3
  rm -rf /tmp/tr_$name
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o t -- "$@" > "$OUT/$name.out" 2> "$OUT/$name.err" ); echo "$name exit $?" | tee -a "$OUT/summary.txt"
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" "$mark" $need > "$OUT/${name}_timeline.txt" 2>&1
  [ -n "$f" ] && python "$ROOT/scripts/step_timeline.py" "$f" x --tail 60 > "$OUT/${name}_tail.txt" 2>&1
  cat "$OUT/${name}_timeline.txt" | tee -a "$OUT/summary.txt"
}
trace headline query_planes_kernel maxsim_pp_kernel python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm
trace cfg5 query_rows_planes maxsim_pp_kernel python "$ROOT/scripts/bench_configs.py" cfg5
trace cfg2 maxsim_stream_kernel "" python "$ROOT/scripts/bench_configs.py" cfg2
trace one maxsim_stream_kernel "" python "$ROOT/scripts/time_one_query.py" 100
for n in cfg5 cfg2 one; do echo "--- $n tail"; cat "$OUT/${n}_tail.txt"; done >> "$OUT/summary.txt"
timeout 300 python - <<'PY' 2>&1 | tee -a "$OUT/summary.txt"
import torch, time
import torch.nn.functional as F
print("torch", torch.__version__)
q = torch.randn(1, 16, 7778, 64, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
fl = 4 * 7778 * 7778 * 1024
base = t(lambda: F.scaled_dot_product_attention(q, k, v)); print("default sdpa ms", base * 1e3, "TF", fl / base / 1e12)
try:
    print("preferred fa lib:", torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    x = t(lambda: F.scaled_dot_product_attention(q, k, v)); print("ck sdpa ms", x * 1e3, "TF", fl / x / 1e12)
    torch.backends.cuda.preferred_rocm_fa_library("aotriton")
except Exception as e:
    print("ck backend:", type(e).__name__, e)
from torch.nn.attention import sdpa_kernel, SDPBackend
for b in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(b):
            x = t(lambda: F.scaled_dot_product_attention(q, k, v)); print(b, "ms", x * 1e3, "TF", fl / x / 1e12)
    except Exception as e:
        print(b, type(e).__name__, str(e)[:200])
# fp16 instead of bf16, and the (B, T, h, d) layout the QKV projection writes
qh, kh, vh = q.half(), k.half(), v.half()
x = t(lambda: F.scaled_dot_product_attention(qh, kh, vh)); print("fp16 ms", x * 1e3, "TF", fl / x / 1e12)
qkv = torch.randn(1, 7778, 3, 16, 64, device="cuda", dtype=torch.bfloat16)
q2, k2, v2 = qkv.permute(2, 0, 3, 1, 4)
x = t(lambda: F.scaled_dot_product_attention(q2, k2, v2)); print("strided qkv ms", x * 1e3, "TF", fl / x / 1e12)
for T in (512, 2048, 4096, 8192):
    qq = torch.randn(max(1, 8192 // T), 16, T, 64, device="cuda", dtype=torch.bfloat16)
    x = t(lambda: F.scaled_dot_product_attention(qq, qq, qq)); print("T", T, "B", qq.shape[0], "ms", x * 1e3, "TF", 4 * qq.shape[0] * T * T * 1024 / x / 1e12)
PY
echo "== $(date) done" | tee -a "$OUT/summary.txt"
