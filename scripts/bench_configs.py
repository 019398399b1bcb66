"""Throughput of the other BASELINE.json configs on one MI355X (parity-test cases, not the bench line).

    python scripts/bench_configs.py [cfg2 cfg3 cfg4 cfg5]   -> one JSON object per config on stdout

Everything is resident in HBM before timing; kernels are launched on torch's current stream and timed with
torch.cuda events (that stream IS the launch stream here).  Algorithmic bytes/flops follow SURVEY.md section 8d.
"""

from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import raglite_amd  # noqa: E402

HBM_PEAK, MFMA_F32_PEAK = 8000.0, 157.3  # GB/s, TFLOP/s (MI355X_MICROARCH.md)


def timed(fn, iters: int, warmup: int = 3) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def cfg2():
    """1 M x 1024 fp32, single-query cosine top-100."""
    n, d, k = 1_000_000, 1024, 100
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=2)
    q = torch.empty((64, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q, seed=20)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    i = [0]

    def one():
        idx.search_rows(q[i[0] % 64], k)
        i[0] += 1

    ms = timed(one, 50)
    ms_scan = idx.time_kernel(1, q[:1], 20) / 20
    # CPU reference on a sample: NumPy sgemv + argpartition over 100k rows, scaled
    Eh = E[:100_000].cpu().numpy()
    qh = q[0].cpu().numpy()
    from oracle import oracle

    t0 = time.perf_counter()
    for _ in range(5):
        oracle.search_rows(Eh, qh, k, "cosine", np.float32)
    cpu = (time.perf_counter() - t0) / 5 * (n / 100_000)
    s, r = idx.search_rows(q[0], k)
    ref = oracle.similarity(Eh, qh, "cosine")
    return {
        "config": "cfg2: 1M x 1024 fp32, B=1 cosine top-100", "queries_per_s": 1e3 / ms, "ms_per_query": ms,
        "scan_kernel_ms": ms_scan, "scan_GBps": 4.0 * n * d / (ms_scan * 1e-3) / 1e9,
        "scan_frac_of_hbm_peak": 4.0 * n * d / (ms_scan * 1e-3) / 1e9 / HBM_PEAK,
        "cpu_numpy_queries_per_s_scaled_from_100k_rows": 1.0 / cpu,
        "sample_score_abs_err": float(np.abs(ref[r.cpu().numpy()[r.cpu().numpy() < 100_000]] -
                                             s.cpu().numpy()[r.cpu().numpy() < 100_000]).max(initial=0.0)),
    }


def cfg3(storage="f32"):
    """ColBERT rerank: 32 query vectors x 256 candidate chunks x 64 vectors/chunk, d = 128."""
    d, nq, n_cand, rows, n_chunks = 128, 32, 256, 64, 16384  # 1 M candidate vectors in the pool
    E = torch.empty((n_chunks * rows, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=3)
    off = np.arange(0, n_chunks * rows + 1, rows, dtype=np.int64)
    idx = raglite_amd.DeviceIndex(E.half() if storage == "f16" else E, off, metric="dot", storage=storage)
    nb = 4096
    Q = torch.empty((nb, nq, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=30)
    cand = torch.randint(0, n_chunks, (nb, n_cand), device="cuda", dtype=torch.int32)
    ms = timed(lambda: idx.maxsim_rerank(Q, cand), 10)
    qps = nb / (ms * 1e-3)
    bytes_q, flops_q = n_cand * rows * d * (2.0 if storage == "f16" else 4.0), 2.0 * nq * n_cand * rows * d
    return {
        "config": f"cfg3: MaxSim rerank 32 x (256 x 64) x 128, 4096 queries per launch, {storage}-stored corpus", "queries_per_s": qps,
        "ms_per_launch": ms, "GBps_algorithmic": qps * bytes_q / 1e9, "frac_of_hbm_peak": qps * bytes_q / 1e9 / HBM_PEAK,
        "TFLOPs_fp32": qps * flops_q / 1e12, "frac_of_mfma_f32_peak": qps * flops_q / 1e12 / MFMA_F32_PEAK,
        "note": "candidates are drawn from a 1M-vector pool (512 MB): partly L2/MALL-resident",
    }


def cfg3_f16():
    return cfg3("f16")


def cfg4():
    """Late-chunking pool + L2-norm + fp16 over 100 k sentences (U{4..60} tokens), d = 1024; adapter matvec."""
    d, S = 1024, 100_000
    rng = np.random.default_rng(4)
    lens = rng.integers(4, 61, size=S)
    ends = np.cumsum(lens)
    begins = ends - lens
    T = int(ends[-1])
    tokens = torch.empty((T, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(tokens, seed=4)
    b = torch.as_tensor(begins, device="cuda")
    e = torch.as_tensor(ends, device="cuda")
    ms = timed(lambda: raglite_amd.pool_norm(tokens, b, e), 10)
    bytes_alg = 4.0 * T * d + 2.0 * S * d
    A = torch.empty((d, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(A, seed=40)
    q1 = torch.empty((1, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q1, seed=41)
    q1000 = torch.empty((1000, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(q1000, seed=42)
    ms_a1 = timed(lambda: raglite_amd.adapter_apply(A, q1), 50)
    ms_a1000 = timed(lambda: raglite_amd.adapter_apply(A, q1000), 5)
    return {
        "config": f"cfg4: pool+norm+fp16, {S} sentences, {T} token rows x 1024", "ms": ms,
        "sentences_per_s": S / (ms * 1e-3), "GBps_algorithmic": bytes_alg / (ms * 1e-3) / 1e9,
        "frac_of_hbm_peak": bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK,
        "adapter_B1_ms": ms_a1, "adapter_B1000_ms": ms_a1000,
    }


def cfg5():
    """Per-GPU part of the 8-GPU config: 1.25 M x 1024 shard, 1000 queries, cosine top-100."""
    n, d, B, k = 1_250_000, 1024, 1000, 100
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=5)
    Q = torch.empty((B, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=50)
    idx = raglite_amd.DeviceIndex(E, metric="cosine")
    ms = timed(lambda: idx.search_rows(Q, k), 3, warmup=1)
    return {
        "config": "cfg5 (one of 8 shards): 1.25M x 1024 fp32, B=1000 cosine top-100", "ms_per_batch": ms,
        "queries_per_s_per_shard_scan": B / (ms * 1e-3), "TFLOPs_fp32": 2.0 * B * n * d / (ms * 1e-3) / 1e12,
        "frac_of_mfma_f32_peak": 2.0 * B * n * d / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK,
    }


if __name__ == "__main__":
    raglite_amd.set_device(0)
    which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4", "cfg5"]
    for name in which:
        print(json.dumps({k: (round(v, 6) if isinstance(v, float) else v) for k, v in globals()[name]().items()}), flush=True)
        torch.cuda.empty_cache()
