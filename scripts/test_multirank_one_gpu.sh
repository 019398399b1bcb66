#!/bin/bash
# Exercise bench.py's N > 1 path (shard by chunk, device exchange + merge, MAX over ranks, rank-0 JSON) on a box with ONE
# GPU: two / four ranks share cuda:0 and the all-gather runs over gloo.  Throughput is meaningless here; the check is
# that the run completes, prints one JSON line with n_gpus = N, and that the merged results equal the single-rank ones
# (tests/test_gpu_sharded.py checks the latter bit for bit).
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 4 --warmup 1 --rows 200000 --same-gpu --backend gloo 2>/dev/null | tail -1 | cut -c1-400
done
