"""End-to-end `embed_strings` on one MI355X with the PyTorch-ROCm token embedder.

    python scripts/bench_embed.py [n_sentences]   -> one JSON object

bge-m3's architecture (XLM-RoBERTa-large: 24 layers, d = 1024, 16 heads, FFN 4096) with random weights and the hashing
tokenizer -- no checkpoint or SentencePiece model can be fetched here -- so the numbers are throughput only.
Reports where the time goes: tokenise + plan (host), encoder forward (PyTorch: hipBLASLt + SDPA), late-chunking pool
(`rl_pool_norm`).
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
from raglite_amd import _embed  # noqa: E402
from raglite_amd._torch_embedder import TorchTokenEmbedder  # noqa: E402


def sentences(n: int, seed: int = 0) -> list[str]:
    """n sentences of 4-19 words drawn from a fixed vocabulary of 20 000 random words of 2-8 letters (the hashing tokenizer stand-in keeps one id
    per distinct piece: unbounded random words would fill its 250 k-entry table at cfg 4's scale)."""
    rng = np.random.default_rng(seed)
    vocab = ["".join(chr(97 + int(c)) for c in rng.integers(0, 26, size=int(rng.integers(2, 9)))) for _ in range(20_000)]
    lens = rng.integers(4, 20, size=n)
    picks = rng.integers(0, len(vocab), size=int(lens.sum()))
    out, at = [], 0
    for ln in lens:
        words = [vocab[i] for i in picks[at : at + ln]]
        at += int(ln)
        out.append(" ".join(words).capitalize() + ". ")
    return out


def run(n: int, pad_queries: bool = True) -> dict:
    raglite_amd.set_device(0)
    emb = TorchTokenEmbedder.bge_m3_shaped(device="cuda")
    cfg = raglite_amd.HotPathConfig()
    sents = sentences(n)
    raglite_amd.embed_strings(sents[:200], config=cfg, embedder=emb)  # warm-up (hipBLASLt heuristics, SDPA)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = raglite_amd.embed_strings(sents, config=cfg, embedder=emb)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    # the same document, stage by stage
    t0 = time.perf_counter()
    counts = _embed.count_sentence_tokens(sents, emb)
    plan = _embed.plan_segments(counts, emb.n_ctx(), emb.n_batch)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    tokens, begins, ends = _embed.plan_document(sents, emb)
    t_enqueue = time.perf_counter() - t0  # (nothing in the loop waits for the device: this is host time)
    torch.cuda.synchronize()
    t_plan_embed = time.perf_counter() - t0
    b = torch.as_tensor(begins, device="cuda")
    e = torch.as_tensor(ends, device="cuda")
    raglite_amd.pool_norm(tokens, b, e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        raglite_amd.pool_norm(tokens, b, e)
    torch.cuda.synchronize()
    t_pool = (time.perf_counter() - t0) / 10
    T = int(tokens.shape[0])
    seg_tokens = [int(counts[a:c].sum()) for a, _, c in plan]
    params = sum(p.numel() for p in emb.encoder.parameters())
    enc_params = params - emb.encoder.tok.weight.numel() - emb.encoder.pos.weight.numel()
    flops_dense = 2.0 * enc_params * sum(seg_tokens)
    flops_attn = sum(4.0 * t * t * emb.shape.hidden * emb.shape.layers for t in seg_tokens)
    t_enc = max(t_plan_embed - t_host, 1e-9)
    return {
        "workload": f"embed_strings end to end: {n} sentences, {T} token rows x 1024, bge-m3-shaped encoder (random weights) -> rl_pool_norm",
        "value": round(T / total, 1), "unit": "token rows/s",
        "sentences": n, "token_rows": T, "segments": len(plan), "out_shape": list(out.shape),
        "embed_strings_s": round(total, 4), "sentences_per_s": round(n / total, 1), "token_rows_per_s": round(T / total, 1),
        "host_tokenise_plan_s": round(t_host, 4), "host_enqueue_s": round(t_enqueue, 4), "plan_plus_encoder_s": round(t_plan_embed, 4),
        "pool_norm_ms": round(t_pool * 1e3, 4),
        "encoder_TFLOPs_bf16": round((flops_dense + flops_attn) / t_enc / 1e12, 2),
        "encoder_dense_share_of_flops": round(flops_dense / (flops_dense + flops_attn), 3),
        "encoder_dense_TFLOPs_bf16": round(flops_dense / t_enc / 1e12, 2),
        "note": "random weights, hashing tokenizer: throughput only; dense = 2 x encoder parameters per token, attention = 4 T^2 d per layer and "
                "segment; both over the encoder's wall time (tokenising the next segment overlaps the GPU)",
    }


def main() -> None:
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)))


if __name__ == "__main__":
    main()
