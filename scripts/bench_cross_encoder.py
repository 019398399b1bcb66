"""Cross-encoder rerank throughput, ms-marco-MiniLM-L-12 shape, random-initialised weights.

    python scripts/bench_cross_encoder.py [--candidates 32] [--doc-tokens 480] [--reps 10] [--cpu-pairs 64]

One "query" = one `rank(query, docs)` call over `--candidates` passages (the reference reranks
oversample * num_results = 4 * 8 = 32 candidates, `src/raglite/_search.py:399-414`), each pair close to the 512-token
limit.  Reported: pairs/s and queries/s on cuda:0 in bf16 and fp32 (tokenisation included -- it is part of `rank`),
and the same module on the host cores in fp32 (the arithmetic FlashRank's ONNX session runs) on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from raglite_amd._cross_encoder import TorchCrossEncoderRanker  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=32)
    ap.add_argument("--doc-tokens", type=int, default=480)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu-pairs", type=int, default=64)
    a = ap.parse_args()
    words = [f"w{i:03d}" for i in range(997)]
    docs = [" ".join(words[(7 * c + 13 * j) % 997] for j in range(a.doc_tokens // 2)) for c in range(a.candidates)]  # word + space = 2 tokens
    query = "which passage mentions " + " ".join(words[3:9])
    out = {"shape": "ms-marco-MiniLM-L-12 (12 x 384, 12 heads, FFN 1536)", "candidates": a.candidates}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        rk = TorchCrossEncoderRanker.minilm_l12_shaped(device="cuda", dtype=dt, pairs_per_batch=256)
        toks = sum(len(p[0]) for p in rk.encode_pairs(query, docs))
        rk.rank(query=query, docs=docs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            rk.rank(query=query, docs=docs)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / a.reps
        t0 = time.perf_counter()
        for _ in range(a.reps):
            rk.encode_pairs(query, docs)
        tok_s = (time.perf_counter() - t0) / a.reps
        out[name] = {"ms_per_query": round(dt_s * 1e3, 3), "pairs_per_s": round(a.candidates / dt_s, 1),
                     "tokens_per_s": round(toks / dt_s), "host_tokenise_ms": round(tok_s * 1e3, 3), "tokens_per_pair": toks // a.candidates}
    cpu = TorchCrossEncoderRanker.minilm_l12_shaped(device="cpu", dtype=torch.float32, pairs_per_batch=16)
    sample = (docs * (a.cpu_pairs // len(docs) + 1))[: a.cpu_pairs]
    cpu.rank(query=query, docs=sample[:4])
    t0 = time.perf_counter()
    cpu.rank(query=query, docs=sample)
    dt_s = time.perf_counter() - t0
    out["cpu_fp32"] = {"pairs_per_s": round(len(sample) / dt_s, 1), "threads": torch.get_num_threads(), "sample_pairs": len(sample)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
