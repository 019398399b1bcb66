"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): MaxSim late-interaction retrieval, 32 query vectors x 1,000,000 chunk
vectors, d = 1024, fp32, exact top-100 chunks -- the shape `metric` is quoted on.  The corpus is
synthetic U(-1,1) (counter-based generator, identical bits on CPU and GPU) grouped into ragged chunks
of 1..15 rows (mean 8, RAGLite's multi-vector chunks), resident in HBM before the timed region.

A step = one batch of QUERIES_PER_STEP queries pushed through the hot path (`rl_maxsim_topk_batch` on
device pointers): the queries' fp16 (hi, lo) fragments once, one corpus pass of the MFMA streaming kernel
per TWO queries (per query with the exact-fp32 arithmetic), one batched exact selection, then the
exchange step (N > 1: ONE RCCL all-gather of every rank's local top-k, (QB, k, 2) int32) and the device
merge.  value = queries / second over the whole job.  (128 queries per step: selection and exchange
are per-step costs, 8 % of a step on a 125 k-row shard at 32 queries per step, 2 % at 128; at N = 1
the rate does not depend on it.)

N > 1: the 1 M-row corpus is sharded by chunk across the ranks (strong scaling on the metric's own
shape); every rank receives the same queries.

Extra objects in the JSON line: `roofline` (dominant kernel, HIP-event timed on its launch stream),
`cpu_baseline` (the NumPy oracle on the host cores, rank 0, N = 1 only), `recall_at_100`.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

N_ROWS, DIM, NQ, TOPK = 1_000_000, 1024, 32, 100
QUERIES_PER_STEP = 128
SEED_CORPUS, SEED_QUERY, SEED_CHUNKS = 6, 60, 600
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def chunk_offsets(n_rows: int) -> np.ndarray:
    """Ragged chunk sizes U{1..15} from the shared counter-based generator (deterministic everywhere)."""
    from oracle.oracle import synth_bits

    sizes = (synth_bits(SEED_CHUNKS, 0, n_rows // 4) >> np.uint64(40)) % np.uint64(15) + np.uint64(1)
    off = np.concatenate(([0], np.cumsum(sizes.astype(np.int64))))
    off = off[off < n_rows]
    return np.concatenate((off, [n_rows])).astype(np.int64)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=N_ROWS, help=argparse.SUPPRESS)  # debugging only
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    # NOT the BASELINE.json configuration (that one is fp32, the default): the fp16-stored index of SURVEY.md 8f-1,
    # reported under its own workload name so it can never be mistaken for the headline number.
    ap.add_argument("--storage", choices=("f32", "f16"), default="f32", help=argparse.SUPPRESS)
    ap.add_argument("--queries-per-step", type=int, default=QUERIES_PER_STEP, help=argparse.SUPPRESS)
    # A/B: the exact fp32 MFMA chain instead of the default fp16 (hi, lo) split of the fp32 operands
    ap.add_argument("--exact-fp32", action="store_true", help=argparse.SUPPRESS)
    # Test hooks for the N > 1 code path on a ONE-GPU box (scripts/test_multirank_one_gpu.sh): every rank uses cuda:0
    # and the collective runs over gloo.  Never used by the driver.
    ap.add_argument("--same-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    qps = args.queries_per_step

    import raglite_amd
    from raglite_amd._sharded import ShardedIndex, shard_bounds_by_chunk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    raglite_amd.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    n_rows = args.rows
    off = chunk_offsets(n_rows)
    c_lo, c_hi = shard_bounds_by_chunk(off, world)[rank]
    r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
    local_off = off[c_lo : c_hi + 1] - off[c_lo]
    # ---- synthetic corpus shard, generated in HBM (element (r, c) is stream element r*DIM + c) ----------
    E = torch.empty((r_hi - r_lo, DIM), dtype=torch.float32, device=dev)
    raglite_amd.synth_fill(E, seed=SEED_CORPUS, start=r_lo * DIM)
    if args.storage == "f16":
        E = E.half()  # the corpus IS these fp16 values (what RAGLite stores, `_embed.py:140`)
    index = raglite_amd.DeviceIndex(E, local_off, metric="dot", storage=args.storage)
    if args.exact_fp32:
        index.set_exact_fp32()
    arithmetic = index.arithmetic
    sharded = ShardedIndex(index, row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=local_off)
    n_batches = 4  # distinct query batches, cycled
    queries = torch.empty((n_batches, qps, NQ, DIM), dtype=torch.float32, device=dev)
    raglite_amd.synth_fill(queries, seed=SEED_QUERY)
    torch.cuda.synchronize()

    def step(i: int):
        return sharded.maxsim_topk_batch(queries[i % n_batches], TOPK)

    def fence() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events on the launch stream (torch's current stream IS the stream every kernel of a step is launched on)
    # bracket the timed region as well: (event time) / (passes) cross-checks the per-launch figure of `roofline`.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        last = step(i)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    region_ms_per_pass = ev0.elapsed_time(ev1) / (args.steps * qps)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_queries = args.steps * qps
    result = {
        "metric": "queries/sec, MaxSim 32x1M d=1024 exact top-100",
        "value": total_queries / elapsed,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        # fp32 data in, fp32 scores out, fp32 accumulation; how the products are formed is `arithmetic`
        "dtype": {"fp32_exact": "f32 (v_mfma_f32_16x16x4_f32)",
                  "f16_split": "f32 operands as exact fp16 hi+lo pairs (22 bits), 3 x v_mfma_f32_16x16x32_f16, f32 accumulate",
                  "f16_stored": "f16 storage, f16 x f16 -> f32 MFMA"}[arithmetic],
        "data": "synthetic",
        "config": {
            "workload": f"maxsim_{NQ}x{n_rows}_d{DIM}_top{TOPK}_ragged_chunks_1to15"
                        + ("" if args.storage == "f32" else "_F16_STORED_CORPUS_not_the_baseline_config"),
            "queries_per_step": qps,
            "n_chunks": int(len(off) - 1),
            "parallelism": f"corpus sharded by chunk over {world} GPU(s); per step one all-gather of local top-k + device merge, no host sync",
        },
    }

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel only -----------------
    # In fp16-split arithmetic `rl_maxsim_topk_batch` scores TWO queries per corpus pass (maxsim_stream2_kernel); the
    # other arithmetics make one pass per query (maxsim_stream_kernel).  Either way one launch = one corpus pass.
    iters = 20
    queries_per_launch, kind, qv = 1, 0, queries[0, 0]
    if arithmetic in ("f16_split", "f16_stored"):
        try:
            index.time_kernel(2, queries[0, :2].reshape(2 * NQ, DIM), 3)
            queries_per_launch, kind, qv = 2, 2, queries[0, :2].reshape(2 * NQ, DIM)
        except Exception:  # noqa: BLE001 - shape outside the pair kernel: one query per pass
            pass
    index.time_kernel(kind, qv, 3)  # warm
    ms = index.time_kernel(kind, qv, iters) / iters
    region_ms_per_pass *= queries_per_launch
    elt = 4.0 if args.storage == "f32" else 2.0
    algo_bytes = elt * (r_hi - r_lo) * DIM  # 4*N*d bytes per corpus pass (2*N*d when fp16-stored)
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    traffic = None
    tf = ROOT / "profiles" / "traffic.json"  # filled from a separate rocprofv3 --pmc pass
    if tf.exists() and args.storage == "f32" and n_rows == N_ROWS and world == 1:
        traffic = json.loads(tf.read_text()).get("maxsim_stream2_bytes_per_launch" if queries_per_launch == 2
                                                 else "maxsim_stream_bytes_per_launch")
    useful_tflops = 2.0 * queries_per_launch * NQ * (r_hi - r_lo) * DIM / (ms * 1e-3) / 1e12
    result["roofline"] = {
        "bound": "hbm",
        "kernel": (("rl::maxsim_stream2_kernel<256, false, true>" if arithmetic == "f16_stored" else "rl::maxsim_stream2_kernel<256, false, false>")
                   if queries_per_launch == 2 else
                   {"fp32_exact": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, false, false>",
                    "f16_split": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, false, true>",
                    "f16_stored": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, true, false>"}[arithmetic]) + " (as rocprofv3 names it)",
        "arithmetic": arithmetic, "queries_per_launch": queries_per_launch,
        "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "kernel_ms": ms, "algorithmic_bytes_per_launch": algo_bytes,
        # HIP events around the whole timed region / corpus passes in it: kernel + its share of selection and exchange
        "timed_region_ms_per_pass": region_ms_per_pass,
        # the second roof: 2*nq*N*d flop per query; SURVEY.md 8d prices it against the 157.3 TF fp32 MFMA peak, which the
        # split arithmetic is not bound by (its products run on the fp16 matrix pipe, 3 MFMAs per exact-fp32-equivalent)
        "useful_tflops": useful_tflops, "useful_tflops_over_fp32_mfma_peak": useful_tflops / 157.3,
    }
    result["config"]["corpus_passes_per_step"] = qps // queries_per_launch

    # ---- recall@100 and CPU baseline: NumPy oracle on the host cores (rank 0, N = 1) --------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle

        E_host = E.float().cpu().numpy()
        q_host = queries[(args.steps - 1) % n_batches].cpu().numpy()
        try:
            from threadpoolctl import threadpool_info

            cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:  # noqa: BLE001
            cores = os.cpu_count() or 1
        oracle.maxsim_topk(E_host[:1000], np.arange(0, 1001, 8), q_host[0], 10, np.float32)  # warm BLAS
        n_cpu = 2
        t0 = time.perf_counter()
        refs = [oracle.maxsim_topk(E_host, off, q_host[b], TOPK, np.float32) for b in range(n_cpu)]
        cpu_s = (time.perf_counter() - t0) / n_cpu
        gpu_scores, gpu_ids = (x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x) for x in last)
        recalls, errs = [], []
        for b in range(n_cpu):
            rs, rc = refs[b]
            recalls.append(len(set(rc.tolist()) & set(gpu_ids[b].tolist())) / TOPK)
            errs.append(float(np.max(np.abs(np.sort(rs)[::-1] - np.sort(gpu_scores[b])[::-1]))))
        result["recall_at_100"] = float(np.mean(recalls))
        result["score_max_abs_err"] = float(np.max(errs))
        result["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "queries/s", "cores": int(cores), "kind": "port",
            "sample": f"{n_cpu} queries of the full workload ({NQ}x{n_rows}x{DIM} fp32, ragged chunks, top-{TOPK}) "
                      f"through oracle/oracle.py (NumPy sgemm + maximum.reduceat + lexsort), host cpu_count={os.cpu_count()}",
        }
    if rank == 0:
        print(json.dumps(result))
    index.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
