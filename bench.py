"""bench.py -- the headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]         (N > 1: launches its own N ranks over 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): MaxSim late-interaction retrieval, 32 query vectors x 1,000,000 chunk
vectors, d = 1024, fp32, exact top-100 chunks -- the shape `metric` is quoted on.  The corpus is
synthetic U(-1,1) (counter-based generator, identical bits on CPU and GPU) grouped into ragged chunks
of 1..15 rows (mean 8, RAGLite's multi-vector chunks), resident in HBM before the timed region.

A step = one batch of QUERIES_PER_STEP queries pushed through the hot path (`rl_maxsim_topk_batch` on
device pointers): the queries' fp16 (hi, lo) fragments once, one corpus pass of the MFMA kernel per SIXTEEN
queries over the index' image of the corpus' hi halves (maxsim_pp_kernel, ONE fp16 product per multiply: q_hi . e_hi), the
batched selection of the approximate scores, the collection of every chunk a rigorous error bound cannot rule out
of the top-k (~180 of 125 k per query), their exact scores on the fp32 matrix pipe (maxsim_pairs_packed_kernel) and the
ranking of those -- the exact top-100 of exactly computed scores; the full-precision passes stand behind a device flag
for corpora the bound does not decide (over the rows by the streaming kernels on an index of rows + HI
image, which is what this workload builds -- lazy images, DESIGN.md 3 -- and over the pre-split image once a batch has
fallen back); one pass per
query with the exact-fp32 arithmetic -- and, N > 1 only, two exchange steps: an RCCL all-gather of every rank's k best
approximate scores and bound, (QB, k + 1) float32, before the candidates are collected (one threshold for all shards), and one of
its exact local top-k, (QB, k, 2) int32, with the device merge.  value = queries / second over the whole job.

N > 1: the 1 M-row corpus is sharded by chunk across the ranks (strong scaling on the metric's own
shape); every rank receives the same queries.

Extra objects in the JSON line (rank 0; all but `roofline` at N = 1 only):
  roofline      dominant kernel, HIP-event timed on its launch stream; `sustained`: the fp16 MFMA rate of this box measured in the same run
  fraction_check / bench_schema  every printed fraction verified to lie in (0, 1] before the line leaves (fraction_violations)
  exact_fp32    the same workload with RL_ARITH_FP32_EXACT (v_mfma_f32_16x16x4_f32 chain), >= 5 timed steps
  f16_stored    the same corpus rounded to and stored as fp16 (the reference's pgvector halfvec), own workload name
  f16_queries   ... and the queries as fp16 values too (what embed_strings returns): the one-product pass is exact, its top-k is the result
  recall_at_100 / score_*  all QUERIES_PER_STEP queries of the last timed step against the fp32 NumPy oracle on the
                full corpus, and against a float64 reference on a >= 50 k-row slab
  cpu_baseline  the NumPy oracle on the host cores (the time of that full-corpus check)
  configs       BASELINE.json configs 1-5 on one GPU (scripts/bench_configs.py), outside the headline's timed region
  raglite_shaped  the headline pipeline on unit-norm fp16-rounded and on clustered corpora (what RAGLite stores): throughput,
                candidates per query, fallback, float64 parity at 1e-4 absolute
  candidates_per_query / fallback_steps  what the bound-filtered pipeline did on the headline's own data
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
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0  # dense fp16 / bf16 MFMA peak (same guide; measured 2495 TF)
MFMA_F32_PEAK_TF = 157.3   # fp32 MFMA / vector peak


def chunk_offsets(n_rows: int) -> np.ndarray:
    """Ragged chunk sizes U{1..15} from the shared counter-based generator (deterministic everywhere)."""
    from oracle.oracle import synth_bits

    sizes = (synth_bits(SEED_CHUNKS, 0, n_rows // 4) >> np.uint64(40)) % np.uint64(15) + np.uint64(1)
    off = np.concatenate(([0], np.cumsum(sizes.astype(np.int64))))
    off = off[off < n_rows]
    return np.concatenate((off, [n_rows])).astype(np.int64)


def blas_threads() -> int:
    try:
        from threadpoolctl import threadpool_info

        return int(max([p.get("num_threads", 1) for p in threadpool_info()] or [1]))
    except Exception:  # noqa: BLE001
        return int(os.cpu_count() or 1)


def vendor_gemm_calibration(dev, n_rows: int, iters: int = 20) -> dict:
    """The external yardstick of the pass kernel's main loop (round-5 review, item 1): the VENDOR's fp16 GEMM (hipBLASLt / rocBLAS through
    `torch.matmul`) on the same box, in the same run, on the same pseudo-random data, at the shapes the pass kernel multiplies --
    [n_rows x 1024] . [1024 x 512] (one pass of sixteen 32-vector queries) and . [1024 x 4096] (a whole 128-query step) -- HIP events on
    torch's current stream (the stream torch launches its GEMM on), `iters` launches each.  The vendor kernel also WRITES its C matrix
    (1 GB / 8 GB in fp16) where the pass kernel reduces it to 125 k chunk maxima per query in registers, so a third, square shape
    (8192^3: C traffic negligible) says what the library's best loop sustains on this box when nothing but the main loop matters.
    torch is the sanctioned owner of device memory; this block is a measurement, not on the product path."""
    import torch

    import raglite_amd

    out: dict = {}

    def rate(m: int, n: int, k: int, out_dtype=None) -> float:
        a32 = torch.empty((m, k), dtype=torch.float32, device=dev)
        raglite_amd.synth_fill(a32, seed=SEED_CORPUS)
        a = a32.half()
        del a32
        b32 = torch.empty((k, n), dtype=torch.float32, device=dev)
        raglite_amd.synth_fill(b32, seed=SEED_QUERY)
        b = b32.half()
        del b32
        c = torch.empty((m, n), dtype=out_dtype or torch.float16, device=dev)

        def mm():
            if out_dtype is None:
                torch.mm(a, b, out=c)
            else:
                torch.mm(a, b, out_dtype=out_dtype, out=c)

        for _ in range(3):
            mm()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            mm()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        del a, b, c
        torch.cuda.empty_cache()
        return ms

    for name, (m, n, k) in (("pass_shape", (n_rows, 16 * NQ, DIM)), ("step_shape", (n_rows, QUERIES_PER_STEP * NQ, DIM)), ("square_8192", (8192, 8192, 8192))):
        try:
            ms = rate(m, n, k)
            out[f"{name}_ms"] = ms
            out[f"{name}_tflops"] = 2.0 * m * n * k / (ms * 1e-3) / 1e12
        except Exception as exc:  # noqa: BLE001 - a diagnostic must not take the bench line down
            out[f"{name}_error"] = f"{type(exc).__name__}: {exc}"
    try:  # fp32 C (what the pass kernel accumulates in; twice the C bytes)
        ms = rate(n_rows, 16 * NQ, DIM, torch.float32)
        out["pass_shape_f32_out_ms"] = ms
        out["pass_shape_f32_out_tflops"] = 2.0 * n_rows * 16 * NQ * DIM / (ms * 1e-3) / 1e12
    except Exception as exc:  # noqa: BLE001 (torch builds without `out_dtype`)
        out["pass_shape_f32_out_error"] = f"{type(exc).__name__}: {exc}"[:200]
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=N_ROWS, help=argparse.SUPPRESS)  # debugging only
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-configs", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-vendor-gemm", action="store_true", help=argparse.SUPPRESS)
    # NOT the BASELINE.json configuration (that one is fp32, the default): the fp16-stored index of SURVEY.md 8f-1,
    # reported under its own workload name so it can never be mistaken for the headline number.
    ap.add_argument("--storage", choices=("f32", "f16"), default="f32", help=argparse.SUPPRESS)
    ap.add_argument("--no-f16", action="store_true", help="skip the fp16-stored block (the reference's storage dtype, own workload name)")
    ap.add_argument("--queries-per-step", type=int, default=QUERIES_PER_STEP, help=argparse.SUPPRESS)
    # A/B: the exact fp32 MFMA chain as the HEADLINE arithmetic (the default run reports it in `exact_fp32` anyway)
    ap.add_argument("--exact-fp32", action="store_true", help=argparse.SUPPRESS)
    # Test hooks for the N > 1 code path on a ONE-GPU box (scripts/test_multirank_one_gpu.sh): every rank uses cuda:0
    # and the collective runs over gloo.  Never used by the driver.
    ap.add_argument("--split", action="store_true", help="also at N = 1: per-rank split of a step into passes / other local work / exchange")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="route option set as a process-wide default before any index exists (A/B runs; see include/raglite_hip.h 'options')")
    ap.add_argument("--same-gpu", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become N ranks (one process per GPU) through torch.distributed.run, exactly the
        # command the driver would have used.  Fewer than N visible devices is an error, never a silent one-GPU run.
        import socket
        import subprocess

        if not args.same_gpu:
            import torch

            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible")
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    qps = args.queries_per_step

    import raglite_amd
    from raglite_amd._sharded import ShardedIndex, shard_bounds_by_chunk

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    raglite_amd.set_device(local_rank)
    for item in args.opt:
        opt_name, opt_value = item.split("=", 1)
        raglite_amd.set_default_option(opt_name, int(opt_value))
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:  # n_gpus below is what the process group says, not what argv says
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()

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
    # N > 1: the exchange step runs behind the C ABI (rl_allgather_merge_topk over librccl); torch.distributed only hands
    # the communicator id around, synchronises the ranks around the timed region and takes the max of their clocks.
    comm = None
    if world > 1 and args.backend == "nccl":
        comm, comm_err = raglite_amd.Communicator.agreed()  # every rank gets one, or none does (tests/test_sharded_gloo8.py)
        if comm is None:
            print(f"[rank {rank}] no RCCL communicator behind the C ABI ({comm_err or 'another rank could not take part'}); "
                  "the exchange step goes through torch.distributed on every rank", file=sys.stderr)
    sharded = ShardedIndex(index, row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=local_off, comm=comm)
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

    def timed_steps(n_steps: int, n_warmup: int, first: int = 0):
        """W untimed steps, then exactly K steps (query batches first, first + 1, ...) bracketed by barrier + synchronize
        on both sides; max over ranks."""
        for i in range(n_warmup):
            step(i)
        fence()
        # HIP events on the launch stream (torch's current stream IS the stream every kernel of a step is launched on)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        out = None
        for i in range(n_steps):
            out = step(first + i)
        ev1.record()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ev0.elapsed_time(ev1), out

    elapsed, region_ms, last = timed_steps(args.steps, args.warmup)

    # How the data decided the bound-filtered pipeline (outside the timed region: reading the counters synchronises).  The steps cycle
    # through n_batches distinct query batches, so one more run of each tells what every timed step did.
    filt = []
    for i in range(min(n_batches, args.steps + args.warmup)):
        step(i)
        filt.append(index.filter_stats())
    fence()
    if filt and filt[0]["kind"] != "none":
        fallback_steps = sum(int(filt[i % len(filt)]["fallback"]) for i in range(args.steps))
        filter_block = {
            "kind": filt[0]["kind"],
            "candidates_per_query": {"mean": float(np.mean([f["candidates_per_query_mean"] for f in filt])),
                                     "max": int(max(f["candidates_per_query_max"] for f in filt))},
            "list_capacity": filt[0]["list_capacity"], "fallback_steps": fallback_steps, "of_steps": args.steps,
            "note": "rank 0's shard; chunks whose approximate score is within 2m of the k-th best, re-scored exactly; fallback = the "
                    "guarded full-precision passes ran (list overflow / unusable bound)",
        }
    else:
        filter_block = {"kind": "none", "fallback_steps": 0, "of_steps": args.steps}

    total_queries = args.steps * qps
    exchange = ("no exchange step at N = 1" if world == 1 else
                f"corpus sharded by chunk over {world} GPUs; per step one all-gather of every rank's k best approximate scores (one candidate threshold for all shards), one of its exact local top-k + device merge, no host sync"
                + (" (rl_allgather_merge_topk: librccl through the C ABI)" if comm is not None else " (torch.distributed)"))
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
                  "f16_split": "f32 in / f32 scores: candidate chunks found with the fp16 hi halves of corpus and queries (1 x v_mfma_f32_16x16x32_f16 per "
                               "multiply, f32 accumulate, rigorous error bound), their scores by v_mfma_f32_16x16x4_f32 (exact f32 products); "
                               "full-precision fallback: 3 x v_mfma_f32_16x16x32_f16 on hi+lo pairs (22 bits)",
                  "f16_stored": "f16 storage, f16 x f16 -> f32 MFMA"}[arithmetic],
        "data": "synthetic",
        "route_options": {"non_default": args.opt, "note": "none = the shipped defaults; set with --opt NAME=VALUE (rl_set_default_option), never from the environment"},
        "candidates_per_query": filter_block.get("candidates_per_query"),
        "fallback_steps": filter_block["fallback_steps"],
        "filter": filter_block,
        "config": {
            "workload": f"maxsim_{NQ}x{n_rows}_d{DIM}_top{TOPK}_ragged_chunks_1to15"
                        + ("" if args.storage == "f32" else "_F16_STORED_CORPUS_not_the_baseline_config"),
            "queries_per_step": qps,
            "n_chunks": int(len(off) - 1),
            "parallelism": exchange,
        },
    }

    mem = index.memory()
    kept = {name: mem[name] for name in ("rows", "presplit_image", "hi_image", "hi_plane")}
    result["index_memory"] = {**kept, "times_corpus": sum(kept.values()) / max(1, kept["rows"]),
                              "note": "rank 0's shard, bytes, right after the timed steps: what an index that serves this workload HAS (lazy images, the "
                                      "default since round 5: an image is built by the first call whose route reads it -- MaxSim batches the HI image; "
                                      "--opt lazy_images=0: every image with the index, 3 x)"}

    result["index_memory_times_corpus"] = result["index_memory"]["times_corpus"]  # (top-level scalar: the driver's record keeps scalars)

    # ---- roofline of the dominant kernel: HIP events on the launch stream, kernel only -----------------------------
    # One launch = one corpus pass of EIGHT queries through maxsim_gemm_kernel (matrix-pipe-bound).  Big fp32 corpora in split
    # arithmetic: the approximate pass over the HI image (kind 6: 1 fp16 MFMA product per multiply; the candidates it leaves
    # are re-scored exactly on the fp32 matrix pipe), else the full-precision pass over the pre-split image (kind 3: 3 products;
    # fp16-stored corpus: 2).  Otherwise two queries or one query (exact fp32) per pass through the HBM-bound streaming kernels.
    iters = 20
    rows_local = r_hi - r_lo
    elt = 4.0 if args.storage == "f32" else 2.0
    algo_bytes_per_pass = elt * rows_local * DIM  # 4*N*d bytes per corpus pass (2*N*d when fp16-stored)
    streamed_bytes = algo_bytes_per_pass
    kind, per_launch = 0, 1
    # (the approximate pass multiplies q_hi.e_hi only -- kind 6; option hi_products = 2: two products -- kind 5)
    # kind 7: SIXTEEN queries per pass through maxsim_pp.hip (the default); option pp_pass = 0: the eight-query pass (kind 6)
    one_product = index.get_option("hi_products") == 1
    no_pp = not index.get_option("pp_pass")
    hi_kind = 5 if not one_product else (6 if no_pp else 7)
    for cand_kind, cand_q in ((hi_kind, 16 if hi_kind == 7 else 8), (6, 8), (3, 8), (2, 2)):
        if cand_kind in (5, 6, 7) and (not index.get_option("hi_maxsim") or (cand_kind == 6 and not one_product)):
            continue
        if arithmetic in ("f16_split", "f16_stored"):
            try:
                index.time_kernel(cand_kind, queries[0, :cand_q].reshape(cand_q * NQ, DIM), 2)
                kind, per_launch = cand_kind, cand_q
                break
            except Exception:  # noqa: BLE001 - that kernel does not apply to this index / shape
                continue
    if kind == 7 and NQ == 32 and qps >= 32:
        # the batch pipeline launches ALL the passes of a step at once (grid row = pass of sixteen queries): the launch that is timed -- and
        # that rocprofv3 sees -- is that one, per_launch queries = per_launch / 16 passes over the HI image
        per_launch = (qps // 16) * 16
    passes_per_launch = per_launch // 16 if kind == 7 else 1
    # two variables on purpose (round 4 printed an HBM fraction of 6.08 by reusing the per-LAUNCH figure for a one-pass kernel):
    # `algo_bytes_per_pass` is one corpus pass, `algo_bytes_per_launch` what the timed launch of `passes_per_launch` passes stands for
    algo_bytes_per_launch = algo_bytes_per_pass * passes_per_launch
    if kind in (5, 6, 7):
        streamed_bytes = 2.0 * rows_local * DIM * passes_per_launch  # the HI image: 2 B per element, once per pass
    qv = queries[0, :per_launch].reshape(per_launch * NQ, DIM)
    index.time_kernel(kind, qv, 3)  # warm
    ms = index.time_kernel(kind, qv, iters) / iters
    fp32_equiv_flops = 2.0 * per_launch * NQ * rows_local * DIM  # 2*nq*N*d per query
    hbm = {"achieved": streamed_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": streamed_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes_per_launch,
           "streamed_bytes_per_launch": streamed_bytes}
    kernel_name = {5: "rl::maxsim_gemm_kernel<2, false, 0, true, false>", 6: "rl::maxsim_gemm_kernel<2, false, 0, true, true>", 7: "rl::maxsim_pp_kernel<0, 0, false>",
                   3: "rl::maxsim_gemm_kernel<2, false, 0, true, false>" if arithmetic == "f16_stored" else "rl::maxsim_gemm_kernel<2, false, 0, false, false>",
                   2: "rl::maxsim_stream2_kernel<256, false, true>" if arithmetic == "f16_stored" else "rl::maxsim_stream2_kernel<256, false, false>",
                   0: {"fp32_exact": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, false, false>",
                       "f16_split": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, false, true>",
                       "f16_stored": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, true, false>"}[arithmetic]}[kind]
    # HBM traffic: FETCH_SIZE of a separate rocprofv3 --pmc pass, taken ONLY from a record filed under the very kernel
    # name this block reports (profiles/traffic.json "by_kernel", written by scripts/summarize_pmc.py --traffic-json) -- else null
    traffic, traffic_source = None, None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists() and args.storage == "f32" and n_rows == N_ROWS and world == 1:
        rec = (json.loads(tf.read_text()).get("by_kernel") or {}).get(kernel_name)
        if rec and rec.get("passes_per_launch") and rec.get("bytes_per_launch"):
            traffic = float(rec["bytes_per_launch"]) / rec["passes_per_launch"] * passes_per_launch
            traffic_source = (f"static: profiles/traffic.json by_kernel[{kernel_name!r}] <- {rec.get('source')} (rocprofv3 --pmc FETCH_SIZE x 2 for the "
                              f"gfx950 half-count, separate run; {rec.get('dispatches')} dispatches of {rec['passes_per_launch']} passes)")
    kernel_name += " (as rocprofv3 names it)"
    if kind in (3, 5, 6, 7):
        # q_hi.e_hi + q_hi.e_lo + q_lo.e_hi: what the split arithmetic needs on the fp16 pipe; an fp16-stored corpus has no e_lo,
        # and the approximate pass over the HI image (kind 5) leaves the e_lo product to the exact re-scoring of its candidates
        products = 1.0 if kind in (6, 7) else 2.0 if (arithmetic == "f16_stored" or kind == 5) else 3.0
        mfma_flops = products * fp32_equiv_flops
        achieved = mfma_flops / (ms * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                              "frac": achieved / MFMA_F16_PEAK_TF, "traffic": traffic,
                              "algorithmic_flops_per_launch": mfma_flops,
                              "flops_note": f"{products:.0f} fp16 MFMA products per fp32-equivalent multiply-add (SURVEY.md 8d: 2*32*N*d per query), {per_launch} queries per launch"
                                            + ("; approximate pass over the HI image, its candidates re-scored exactly by maxsim_pairs_kernel inside the timed step" if kind in (5, 6, 7) else ""),
                              "hbm": hbm}
        # The same fraction against what the matrix pipe SUSTAINS on this box in this run: the pass kernel's MFMA stream alone (8 waves per CU,
        # 128 accumulators each, register-resident pseudo-random fp16 operands; no loads, no LDS, no epilogue -- rl_time_kernel kind 9).
        # The nominal 2.5 PFLOP/s is 16 cycles per MFMA at 2.4 GHz; under a full MFMA load the shader clock settles well below that.
        rl_block = result["roofline"]
        try:
            index.time_kernel(9, qv, 3)
            rate_ms = index.time_kernel(9, qv, iters) / iters
            n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
            rate_tf = n_cu * 8 * 1000 * 32 * (2.0 * 16 * 16 * 32) / (rate_ms * 1e-3) / 1e12
            # SCALARS of the roofline block itself (the driver's record keeps scalars only: round 5's nested `sustained` never reached it)
            rl_block.update({
                "sustained_tflops": rate_tf, "frac_of_sustained": achieved / rate_tf, "sustained_frac_of_nominal_peak": rate_tf / MFMA_F16_PEAK_TF,
                "shader_clock_ghz": rate_tf / MFMA_F16_PEAK_TF * 2.4, "sustained_kernel_ms": rate_ms,
                "sustained_how": f"rl::mfma_f16_rate_kernel, same run, right after the pass kernel: {iters} launches of {n_cu} workgroups x 8 waves x 32 000 "
                                 "v_mfma_f32_16x16x32_f16 on register-resident pseudo-random operands (HIP events); `frac` stays against the nominal peak"})
        except Exception as exc:  # noqa: BLE001 - a diagnostic must not take the bench line down
            rl_block["sustained_error"] = f"{type(exc).__name__}: {exc}"
        # ... and against the vendor's GEMM on the same box in the same run (vendor_gemm_calibration): the yardstick the main loop is held to
        if rank == 0 and world == 1 and kind == 7 and not args.no_vendor_gemm:
            try:
                vg = vendor_gemm_calibration(dev, rows_local, iters)
            except Exception as exc:  # noqa: BLE001
                vg = {"error": f"{type(exc).__name__}: {exc}"}
            for key, val in vg.items():
                rl_block[f"vendor_gemm_{key}"] = val
            best_like = max((vg.get(f"{n}_tflops") or 0.0) for n in ("pass_shape", "step_shape", "pass_shape_f32_out"))
            if best_like > 0.0:
                rl_block["vendor_gemm_tflops"] = best_like  # the vendor's best on the pass kernel's own shapes
                rl_block["pass_kernel_over_vendor"] = achieved / best_like  # whole pass kernel (main loop + tile epilogues + flush) / vendor GEMM (main loop + C store)
            if vg.get("square_8192_tflops"):
                rl_block["pass_kernel_over_vendor_square"] = achieved / vg["square_8192_tflops"]
            rl_block["vendor_gemm_how"] = (f"torch.mm (hipBLASLt / rocBLAS) fp16 x fp16 -> fp16 (fp32 accumulate), {iters} launches each, HIP events, same run, same "
                                           f"pseudo-random data: [{rows_local} x {DIM}] . [{DIM} x {16 * NQ}] (one pass), . [{DIM} x {QUERIES_PER_STEP * NQ}] (a step), 8192^3")
        elif rank == 0 and world == 1 and kind == 7:
            rl_block["vendor_gemm_error"] = "skipped: --no-vendor-gemm"
    else:
        result["roofline"] = {"bound": "hbm", **hbm, "traffic": traffic}
    if kind in (5, 6, 7):  # for reference: the full-precision pass (eight queries) the approximate one replaces (and falls back to)
        qv8 = queries[0, :8].reshape(8 * NQ, DIM)
        try:
            index.time_kernel(3, qv8, 2)
            result["roofline"]["full_precision_pass_ms"] = index.time_kernel(3, qv8, iters) / iters
        except raglite_amd._abi.UnsupportedError:  # --opt keep_image=0: no pre-split image, the fallback is the streaming kernel over the rows
            result["roofline"]["full_precision_pass_ms"] = None
    result["roofline"].update({
        "kernel": kernel_name, "arithmetic": arithmetic, "queries_per_launch": per_launch, "passes_per_launch": passes_per_launch, "kernel_ms": ms,
        "kernel_ms_per_pass": ms / passes_per_launch,
        "traffic_source": traffic_source, "algorithmic_bytes_per_launch": algo_bytes_per_launch, "algorithmic_bytes_per_pass": algo_bytes_per_pass,
        # HIP events around the whole timed region / corpus passes in it: the kernel + its share of query split and selection
        "timed_region_ms_per_launch": region_ms / (args.steps * qps) * per_launch,
        "fp32_equivalent_tflops": fp32_equiv_flops / (ms * 1e-3) / 1e12,
        "fp32_equivalent_tflops_over_fp32_mfma_peak": fp32_equiv_flops / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
    })
    result["config"]["corpus_passes_per_step"] = -(-qps // per_launch) * passes_per_launch
    result["config"]["launches_per_step"] = -(-qps // per_launch)

    # ---- where a rank's step goes (N > 1, or --split): passes / the rest of its local work / the exchanges ----------------------------
    # Outside the timed region, the stages of ShardedIndex.maxsim_topk_batch's staged path issued one by one with HIP events between them
    # on the launch stream: begin (query images, the approximate passes, their top-k, this shard's bound) | all-gather of the shards'
    # approximate lists | finish (threshold, collection, exact re-scoring, ranking) | all-gather + merge of the local top-k.  `pass_ms` is
    # passes per step x this rank's kernel time (the live figure of the roofline block), so `local_other_ms` = begin + finish - passes is
    # the part that does not shrink with the shard.  Every rank reports; rank 0 prints all of them, so the first run on real links explains itself.
    if (world > 1 or args.split) and hasattr(index, "maxsim_batch_begin") and kind in (5, 6, 7):
        n_split = max(3, min(args.steps, 10))
        acc_ms = np.zeros(4)
        # A diagnostic must not take the bench line down -- and must not hang it either: the two LOCAL stages run under try, the two
        # COLLECTIVES are always entered (with empty lists once a local stage has failed on this rank), the error is reported afterwards.
        split_err = None
        for it in range(n_split + 1):
            q_b = queries[it % n_batches]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            ev[0].record()
            approx = None
            if split_err is None:
                try:
                    approx = index.maxsim_batch_begin(q_b, TOPK)
                except Exception as exc:  # noqa: BLE001
                    split_err = exc
            if approx is None:
                approx = torch.full((qps, TOPK + 1), float("-inf"), dtype=torch.float32, device=dev)
            ev[1].record()
            all_approx = sharded._allgather_int(approx.contiguous().view(torch.int32)).view(torch.float32)  # noqa: SLF001
            ev[2].record()
            s_loc = c_loc = None
            if split_err is None:
                try:
                    s_loc, c_loc = index.maxsim_batch_finish(q_b, all_approx, rank, TOPK)
                except Exception as exc:  # noqa: BLE001
                    split_err = exc
            if s_loc is None:
                s_loc = torch.full((qps, TOPK), float("-inf"), dtype=torch.float32, device=dev)
                c_loc = torch.full((qps, TOPK), -1, dtype=torch.int32, device=dev)
            ev[3].record()
            sharded._exchange_merge_device(s_loc, c_loc, c_lo, TOPK)  # noqa: SLF001
            ev[4].record()
            fence()
            if it > 0:  # (the first round warms the staged path up)
                acc_ms += np.array([ev[j].elapsed_time(ev[j + 1]) for j in range(4)])
        if split_err is None:
            begin_ms, gather_ms, finish_ms, merge_ms = (acc_ms / n_split).tolist()
            pass_ms = -(-qps // per_launch) * ms
            mine = {"rank": rank, "rows": int(r_hi - r_lo), "step_ms": begin_ms + gather_ms + finish_ms + merge_ms, "pass_ms": pass_ms,
                    "local_other_ms": begin_ms + finish_ms - pass_ms, "exchange_ms": gather_ms + merge_ms,
                    "stages_ms": {"begin": begin_ms, "allgather_approx": gather_ms, "finish": finish_ms, "allgather_merge_topk": merge_ms}}
        else:
            mine = {"rank": rank, "error": f"{type(split_err).__name__}: {split_err}"}
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        result["rank_split"] = {"ranks": everyone, "steps": n_split,
                                "note": "stages issued one by one with HIP events between them (outside the timed region); pass_ms = corpus passes per step x "
                                        "this rank's kernel time; local_other_ms = begin + finish - pass_ms; exchange_ms = the two all-gathers (+ device merge)"}

    single = rank == 0 and world == 1
    # ---- the same workload in exact fp32 arithmetic (the reference's own number format), driver-timed ------------------
    if single and arithmetic == "f16_split" and not args.exact_fp32:
        index.set_exact_fp32(True)
        ex_steps = max(5, min(args.steps, 10))
        ex_first = (args.steps - ex_steps) % n_batches  # so that its last step scores the headline's last query batch
        ex_elapsed, _, ex_last = timed_steps(ex_steps, 2, ex_first)
        index.time_kernel(0, queries[0, 0], 3)
        ex_ms = index.time_kernel(0, queries[0, 0], iters) / iters
        result["exact_fp32"] = {
            "value": ex_steps * qps / ex_elapsed, "unit": "queries/s", "steps": ex_steps, "ms_per_step": 1e3 * ex_elapsed / ex_steps,
            "arithmetic": index.arithmetic, "kernel": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, false, false>",
            "kernel_ms": ex_ms, "queries_per_launch": 1,
            "algorithmic_bytes_per_launch": algo_bytes_per_pass,  # ONE query per corpus pass, one pass per launch
            "frac": algo_bytes_per_pass / (ex_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
            "fp32_mfma_tflops": 2.0 * NQ * rows_local * DIM / (ex_ms * 1e-3) / 1e12,
            "fp32_mfma_frac": 2.0 * NQ * rows_local * DIM / (ex_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
        }
        exact_last = tuple(x.clone() for x in ex_last)
        index.set_exact_fp32(False)  # back to the default arithmetic (rebuilds the corpus image)
        assert index.arithmetic == arithmetic
    else:
        exact_last = None

    # ---- ONE user query at a time: how the reference calls the reranker (`_search.py:394-396`) -- rl_maxsim_topk per query, host loop ------
    # Round 6: one or two queries rank from the row-major fp16 HI plane (2 B per element, HBM-bound streaming kernel) and re-score their
    # candidates exactly over the rows (option hi_few; `rows_route`: the streaming kernel over the fp32 rows it replaces, same results).
    if single and arithmetic == "f16_split" and not args.exact_fp32 and hasattr(index, "maxsim_topk"):
        n_one = min(64, qps)
        qs1 = queries[(args.steps - 1) % n_batches][:n_one]

        def one_by_one():
            outs = [index.maxsim_topk(qs1[i], TOPK) for i in range(n_one)]
            return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])

        def timed_loop():
            for i in range(3):
                index.maxsim_topk(qs1[i], TOPK)
            fence()
            t0 = time.perf_counter()
            got = one_by_one()
            fence()
            return time.perf_counter() - t0, got

        one_s, (o_s, o_c) = timed_loop()
        one_stats = index.filter_stats()
        block = {"workload": f"maxsim_{NQ}x{n_rows}_d{DIM}_top{TOPK}_ONE_QUERY_PER_CALL", "value": n_one / one_s, "unit": "queries/s", "queries": n_one,
                 "ms_per_query": 1e3 * one_s / n_one, "route": one_stats["kind"], "fallback": one_stats["fallback"],
                 "candidates_per_query": {"mean": one_stats["candidates_per_query_mean"], "max": one_stats["candidates_per_query_max"]}}
        if last is not None:  # the same queries went through the batch pipeline in the last timed step: same chunks, same exactly re-scored scores
            b_s, b_c = last[0][:n_one], last[1][:n_one]
            block["chunks_identical_to_batch"] = bool(torch.equal(torch.sort(o_c, dim=1).values, torch.sort(b_c.to(o_c.dtype), dim=1).values))
            block["score_max_abs_diff_vs_batch"] = float((torch.sort(o_s, dim=1).values - torch.sort(b_s, dim=1).values).abs().max())
        try:
            index.time_kernel(10, qs1[0], 3)
            k_ms = index.time_kernel(10, qs1[0], iters) / iters
            plane_bytes = 2.0 * rows_local * DIM
            block.update({"kernel": "rl::maxsim_stream_kernel<256, 2, 0, false, 6, true, false>", "kernel_ms": k_ms, "bound": "hbm",
                          "narrower_image": "row-major fp16 HI plane, 2 B per element (+ 0.5 x the corpus resident)",
                          "streamed_bytes_per_launch": plane_bytes, "frac": plane_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "frac_vs_4B_per_element_whole_query": algo_bytes_per_pass / (one_s / n_one) / 1e9 / HBM_PEAK_GBS})
        except Exception as exc:  # noqa: BLE001
            block["kernel_error"] = f"{type(exc).__name__}: {exc}"
        with index.options(hi_few=0):
            rows_s, (r_s, r_c) = timed_loop()
        block["rows_route"] = {"value": n_one / rows_s, "ms_per_query": 1e3 * rows_s / n_one,
                               "chunks_identical": bool(torch.equal(torch.sort(o_c, dim=1).values, torch.sort(r_c, dim=1).values)),
                               "note": "--opt hi_few=0: the streaming kernel over the fp32 rows (4 B per element), what rounds 1-5 ran"}
        result["one_query"] = block
        result["one_query_value"] = block["value"]

    # ---- the reference's real storage dtype (pgvector halfvec, `_typing.py:211-232`; `_embed.py:140` casts to fp16): the same
    # workload over the SAME corpus rounded to fp16, under its own workload name -- not the BASELINE config, driver-timed ----------
    if single and args.storage == "f32" and not args.exact_fp32 and not args.no_f16:
        E16 = E.half()
        idx16 = raglite_amd.DeviceIndex(E16, local_off, metric="dot", storage="f16")

        def step16(i: int):
            return idx16.maxsim_topk_batch(queries[i % n_batches], TOPK)

        for i in range(2):
            step16(i)
        fence()
        f_steps = max(5, min(args.steps, 10))
        t0 = time.perf_counter()
        for i in range(f_steps):
            last16 = step16(i)
        fence()
        f_elapsed = time.perf_counter() - t0
        f_stats = idx16.filter_stats()  # what the last (untimed-region) step did: the bound-filtered pipeline, or none
        try:  # sixteen queries per pass at one product per multiply (q_hi.e; the candidates are re-scored over the stored rows)
            qv16 = queries[0, :16].reshape(16 * NQ, DIM)
            idx16.time_kernel(7, qv16, 3)
            f_ms = idx16.time_kernel(7, qv16, iters) / iters
            f_kernel, f_per, f_products = "rl::maxsim_pp_kernel<0, 0, false>", 16, 1.0
        except ValueError:  # (no image for the approximate pass: the eight-query kernel at two products, q_hi.e + q_lo.e)
            try:
                qv8 = queries[0, :8].reshape(8 * NQ, DIM)
                idx16.time_kernel(3, qv8, 3)
                f_ms = idx16.time_kernel(3, qv8, iters) / iters
                f_kernel, f_per, f_products = "rl::maxsim_gemm_kernel<2, false, 0, true, false>", 8, 2.0
            except ValueError:  # (--opt keep_image=0: an fp16-stored index has no image at all, its batches stream the stored rows)
                f_ms, f_kernel, f_per, f_products = float("nan"), "none (no image: --opt keep_image=0)", 1, 0.0
        f_flops = f_products * 2.0 * f_per * NQ * rows_local * DIM
        # spot check: query 0 of the last batch against the fp32 NumPy oracle over the stored (fp16) values, first 50 k rows
        result["f16_stored"] = {
            "workload": "maxsim_32x1000000_d1024_top100_F16_STORED_CORPUS_not_the_baseline_config",
            "value": f_steps * qps / f_elapsed, "unit": "queries/s", "steps": f_steps, "ms_per_step": 1e3 * f_elapsed / f_steps,
            "arithmetic": idx16.arithmetic, "kernel": f_kernel, "kernel_ms": f_ms, "products_per_multiply": f_products,
            "candidates_per_query": {"mean": f_stats["candidates_per_query_mean"], "max": f_stats["candidates_per_query_max"]},
            "filter": f_stats["kind"], "fallback": f_stats["fallback"],
            "queries_per_launch": f_per, "bound": "mfma", "achieved_tflops": f_flops / (f_ms * 1e-3) / 1e12,
            "frac": f_flops / (f_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF,
            "hbm_frac": 2.0 * rows_local * DIM / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        f16_last = tuple(x.clone() for x in last16)
        f16_index, f16_matrix = idx16, E16
        # ---- ... and with the QUERIES as fp16 values too (what embed_strings and the query adapter hand over, `_embed.py:140`,
        # `_search.py:62`): rl_maxsim_topk_batch_f16 -- the one-product pass is exact there, its top-k is the result (no bound, no
        # candidate list, no re-scoring kernel).  Own workload name, driver-timed like the block above.
        q16 = queries.half()

        def step16h(i: int):
            return idx16.maxsim_topk_batch(q16[i % n_batches], TOPK)

        for i in range(2):
            step16h(i)
        fence()
        t0 = time.perf_counter()
        for i in range(f_steps):
            last16h = step16h(i)
        fence()
        h_elapsed = time.perf_counter() - t0
        h_stats = idx16.filter_stats()
        result["f16_queries"] = {
            "workload": "maxsim_32x1000000_d1024_top100_F16_STORED_CORPUS_F16_QUERIES_not_the_baseline_config",
            "value": f_steps * qps / h_elapsed, "unit": "queries/s", "steps": f_steps, "ms_per_step": 1e3 * h_elapsed / f_steps,
            "route": h_stats["kind"], "candidates_per_query": {"mean": h_stats["candidates_per_query_mean"], "max": h_stats["candidates_per_query_max"]},
            "fallback": h_stats["fallback"], "kernel": f_kernel, "kernel_ms": f_ms,
            "note": "fp16 x fp16 products are exact in fp32 and neither side has a dropped half: the pass's exact top-k is returned as is",
        }
        result["f16_queries_value"] = result["f16_queries"]["value"]  # (top-level scalar: the driver's record keeps scalars)
        del last16h
    else:
        f16_last = None

    # ---- recall@100, score error and CPU baseline: NumPy oracle on the host cores (rank 0, N = 1) -------------------
    if single and not args.no_cpu_baseline:
        from oracle import oracle

        E_host = E.float().cpu().numpy()
        q_host = queries[(args.steps - 1) % n_batches].cpu().numpy()  # the batch of the last timed step
        cores = blas_threads()
        oracle.maxsim_topk(E_host[:1000], np.arange(0, 1001, 8), q_host[0], 10, np.float32)  # warm BLAS
        t0 = time.perf_counter()
        ref_scores = oracle.maxsim_scores_batch(E_host, off, q_host, np.float32)  # (qps, n_chunks), fp32 as computed
        refs = [oracle.topk_desc(ref_scores[b], TOPK) for b in range(qps)]
        cpu_s = time.perf_counter() - t0
        gpu_scores, gpu_ids = (x.cpu().numpy() for x in last)

        def against(ids, scores):
            rec, err = [], []
            for b in range(qps):
                rs, rc = refs[b]
                rec.append(len(set(rc.tolist()) & set(ids[b].tolist())) / TOPK)
                err.append(float(np.max(np.abs(rs - scores[b]))))  # both sorted by (score desc, id asc)
            return rec, err

        recalls, errs = against(gpu_ids, gpu_scores)
        result["recall_at_100"] = float(np.mean(recalls))
        result["recall_at_100_min"] = float(np.min(recalls))
        result["queries_checked"] = qps
        result["score_max_abs_err"] = float(np.max(errs))  # vs the fp32 NumPy oracle (OpenBLAS sgemm), scores ~ 500
        result["score_scale"] = float(np.abs(gpu_scores).max())
        if exact_last is not None:  # the exact-fp32 run's last step scored the same query batch
            r2, e2 = against(exact_last[1].cpu().numpy(), exact_last[0].cpu().numpy())
            result["exact_fp32"].update({"recall_at_100": float(np.mean(r2)), "score_max_abs_err": float(np.max(e2))})
        # float64 reference on a slab: every one of the slab's top-2048 chunk scores of every query, relative error
        c1 = int(np.searchsorted(off, 50_000, side="left"))
        r1 = int(off[c1])
        slab = raglite_amd.DeviceIndex(E[:r1], off[: c1 + 1], metric="dot", storage=args.storage)
        ks = min(2048, c1)
        ss, sc = (x.cpu().numpy() for x in slab.maxsim_topk_batch(queries[(args.steps - 1) % n_batches], ks))
        slab.close()
        ref64 = oracle.maxsim_scores_batch(E_host[:r1], off[: c1 + 1], q_host, np.float64)
        got64 = np.take_along_axis(ref64, sc.astype(np.int64), axis=1)
        result["score_max_rel_err"] = float(np.max(np.abs(ss - got64) / np.maximum(np.abs(got64), 1e-30)))
        result["score_max_abs_err_vs_f64_slab"] = float(np.max(np.abs(ss - got64)))
        result["score_check"] = (f"{qps} queries x top-{ks} of the first {c1} chunks ({r1} rows) against float64; "
                                 f"recall and score_max_abs_err: all {qps} queries of the last timed step, full corpus, fp32 NumPy oracle")
        if f16_last is not None:  # the fp16-stored run: same slab check over the STORED (fp16) values
            slab16 = raglite_amd.DeviceIndex(f16_matrix[:r1], off[: c1 + 1], metric="dot", storage="f16")
            s16, c16 = (x.cpu().numpy() for x in slab16.maxsim_topk_batch(queries[(args.steps - 1) % n_batches], ks))
            slab16b = slab16
            ref16 = oracle.maxsim_scores_batch(f16_matrix[:r1].float().cpu().numpy(), off[: c1 + 1], q_host, np.float64)
            got16 = np.take_along_axis(ref16, c16.astype(np.int64), axis=1)
            top16 = np.argsort(-ref16, axis=1, kind="stable")[:, :TOPK]
            result["f16_stored"].update({
                "score_max_rel_err": float(np.max(np.abs(s16 - got16) / np.maximum(np.abs(got16), 1e-30))),
                "recall_at_100_slab": float(np.mean([len(set(top16[b].tolist()) & set(c16[b][:TOPK].tolist())) / TOPK for b in range(qps)])),
                "check": f"{qps} queries x top-{ks} of the first {c1} chunks against float64 over the stored fp16 values",
            })
            # fp16 queries: the same slab check against float64 over the fp16 values of BOTH sides
            q16_slab = queries[(args.steps - 1) % n_batches].half()
            s16h, c16h = (x.cpu().numpy() for x in slab16b.maxsim_topk_batch(q16_slab, ks))
            route16h = slab16b.filter_stats()["kind"]
            ref16h = oracle.maxsim_scores_batch(f16_matrix[:r1].float().cpu().numpy(), off[: c1 + 1], q16_slab.float().cpu().numpy(), np.float64)
            got16h = np.take_along_axis(ref16h, c16h.astype(np.int64), axis=1)
            top16h = np.argsort(-ref16h, axis=1, kind="stable")[:, :TOPK]
            result["f16_queries"].update({
                "score_max_rel_err": float(np.max(np.abs(s16h - got16h) / np.maximum(np.abs(got16h), 1e-30))),
                "recall_at_100_slab": float(np.mean([len(set(top16h[b].tolist()) & set(c16h[b][:TOPK].tolist())) / TOPK for b in range(qps)])),
                "check": f"{qps} queries x top-{ks} of the first {c1} chunks against float64 over the fp16 values of corpus and queries (route on the slab: {route16h})",
            })
            slab16b.close()
            del ref16, ref16h
            f16_index.close()
        result["cpu_baseline"] = {
            "value": qps / cpu_s, "unit": "queries/s", "cores": int(cores), "blas_threads": int(cores), "kind": "port",
            "sample": f"{qps} queries (one step) of the full workload ({NQ}x{n_rows}x{DIM} fp32, ragged chunks, top-{TOPK}) through "
                      f"oracle/oracle.py:maxsim_scores_batch (one OpenBLAS sgemm per 65536-row slab for all queries, "
                      f"maximum.reduceat, lexsort top-k); {cpu_s:.1f} s on host cpu_count={os.cpu_count()}, BLAS threads={cores}",
        }
        del E_host, ref_scores, ref64

    # ---- BASELINE.json configs 1-5 on this GPU, outside the headline's timed region -----------------------------------
    if single and not args.no_configs and args.storage == "f32" and n_rows == N_ROWS:
        index.close()
        del E, queries, index, sharded
        torch.cuda.empty_cache()
        sys.path.insert(0, str(ROOT / "scripts"))
        import bench_configs

        result["configs"] = {}
        for name in ("cfg1", "cfg2", "cfg3", "cfg3_pool2g", "cfg4", "cfg4_end_to_end", "cfg5", "cfg5_full_one_gpu", "wide_dims", "beyond_shape"):
            try:
                result["configs"][name] = bench_configs.run(name)
            except Exception as exc:  # noqa: BLE001 - a failing side config must not hide the headline
                result["configs"][name] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        # the headline pipeline on RAGLite-shaped corpora (unit-norm fp16-rounded rows; clustered): own workload names
        result["raglite_shaped"] = {}
        for name in ("shaped_unit", "shaped_clustered"):
            try:
                result["raglite_shaped"][name] = bench_configs.run(name)
            except Exception as exc:  # noqa: BLE001
                result["raglite_shaped"][name] = {"error": f"{type(exc).__name__}: {exc}"}
            torch.cuda.empty_cache()
        print(_finish(result))
        return
    if rank == 0:
        print(_finish(result))
    index.close()
    if world > 1:
        dist.destroy_process_group()


BENCH_SCHEMA = 6  # round 6: sustained / vendor-GEMM yardsticks as SCALARS of `roofline`, index_memory_times_corpus and f16_queries_value at top level; round 5: per-pass / per-launch byte counts kept apart, every printed fraction checked by fraction_violations()


def fraction_violations(node, path: str = "") -> list[str]:
    """Every printed fraction of a roofline must lie in (0, 1]; every HBM-bound block's algorithmic bytes / kernel time must stay
    below the HBM peak unless the block NAMES the narrower image its kernel streams (`narrower_image`).  Returns the violations
    (tests/test_bench_line.py runs this over every recorded line of this schema; main() prints the list in `fraction_check`)."""
    bad: list[str] = []
    if isinstance(node, dict):
        narrower = bool(node.get("narrower_image"))
        for key, val in node.items():
            here = f"{path}.{key}" if path else key
            is_frac = key == "frac" or key.endswith("_frac") or key.startswith("frac_")
            if is_frac and isinstance(val, (int, float)) and not isinstance(val, bool):
                allowed_above_one = narrower and key == "frac_vs_4B_per_element_whole_query"
                if not (val == val and val > 0.0 and (val <= 1.0 or allowed_above_one)):
                    bad.append(f"{here} = {val!r} is not in (0, 1]")
            bad += fraction_violations(val, here)
        if node.get("bound") == "hbm" and not narrower:
            nbytes = node.get("algorithmic_bytes_per_launch", node.get("algorithmic_bytes"))
            ms = node.get("kernel_ms")
            if isinstance(nbytes, (int, float)) and isinstance(ms, (int, float)) and ms > 0 and nbytes / (ms * 1e-3) / 1e9 > HBM_PEAK_GBS:
                bad.append(f"{path or '<root>'}: algorithmic bytes / kernel_ms = {nbytes / (ms * 1e-3) / 1e9:.0f} GB/s exceeds the HBM peak and the block names no narrower image")
    elif isinstance(node, list):
        for i, val in enumerate(node):
            bad += fraction_violations(val, f"{path}[{i}]")
    return bad


def _finish(result: dict) -> str:
    _add_score_tolerance(result)
    result["bench_schema"] = BENCH_SCHEMA
    bad = fraction_violations(result)
    result["fraction_check"] = "ok: every frac / *_frac in (0, 1]" if not bad else {"violations": bad}
    return json.dumps(result)


def _add_score_tolerance(result: dict) -> None:
    """north_star states the bar as "cosine / MaxSim scores within 1e-4 fp32": an ABSOLUTE figure on scores of unit-norm embeddings,
    where a 32-vector MaxSim score is at most 32.  The BASELINE corpus is U(-1, 1)^1024 (rows of norm ~18.5, scores ~755), so the same bar is
    stated here the way it scales: relative to the score, 1e-4 / 32.  Both readings are put next to the measured errors -- the literal
    absolute one on the unit-norm corpus of the `raglite_shaped` block, the relative one on the headline corpus."""
    bar_rel = 1e-4 / NQ
    block = {"north_star": "scores within 1e-4 (fp32) of the reference on cosine / unit-norm MaxSim scores (<= 32 per 32-vector query)",
             "kind": "relative", "bar_rel": bar_rel, "bar_abs_on_unit_norm": 1e-4}
    rel = result.get("score_max_rel_err")
    if rel is not None:
        block.update({"measured_rel_vs_f64": rel, "measured_abs_vs_fp32_oracle": result.get("score_max_abs_err"),
                      "score_scale": result.get("score_scale"), "pass_rel": bool(rel <= bar_rel)})
    unit = (result.get("raglite_shaped") or {}).get("shaped_unit") or {}
    abs_unit = (unit.get("check") or {}).get("score_max_abs_err_vs_f64")
    if abs_unit is not None:
        block.update({"abs_on_unit_norm": abs_unit, "pass_abs_on_unit_norm": bool(abs_unit <= 1e-4)})
    result["score_tolerance"] = block


if __name__ == "__main__":
    main()
