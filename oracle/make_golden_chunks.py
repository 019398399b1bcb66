"""Generate `tests/golden/split_chunks.npz` by running the REFERENCE's own `split_chunks` -- TEST INFRASTRUCTURE.
Run in the authoring container only (needs /root/reference):

    python -m oracle.make_golden_chunks

`src/raglite/_split_chunks.py` imports cleanly here (numpy + scipy only), so the real function runs; its MILP cost
vector -- the partition similarities of `:54-86`, after the heading adjustments -- is captured by wrapping the
`linprog` name inside that module.
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

REFERENCE_SRC = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "split_chunks.npz"


def make_document(rng: np.random.Generator, n: int, dim: int, heading_every: int = 0, common: float = 0.7):
    """n chunklets (sentence-like strings of varied length, optional Markdown headings) + fp16 embeddings that share
    a common direction (the 'discourse vector' real embeddings have)."""
    words = ["alpha", "beta", "gamma", "delta", "retrieval", "augmented", "generation", "kernel", "wavefront", "of", "a", "the"]
    chunklets = []
    for i in range(n):
        if heading_every and i % heading_every == 0:
            chunklets.append(f"# Section {i // heading_every}\n\n")
            continue
        k = int(rng.integers(3, 40))
        chunklets.append(" ".join(words[int(j)] for j in rng.integers(0, len(words), size=k)).capitalize() + ". ")
    base = rng.standard_normal(dim)
    X = common * base[None, :] + rng.standard_normal((n, dim))
    topic = np.cumsum(rng.standard_normal((n, dim)) * 0.35, axis=0)  # slowly drifting topic: neighbours are similar
    X = X + topic
    return chunklets, X.astype(np.float16)


def main() -> None:
    pkg = types.ModuleType("raglite")
    pkg.__path__ = [str(REFERENCE_SRC / "raglite")]
    sys.modules["raglite"] = pkg
    from raglite import _split_chunks as ref  # REAL reference module

    captured: list[np.ndarray] = []
    real_linprog = ref.linprog

    def spy(c, *a, **k):  # noqa: ANN001,ANN002,ANN003
        captured.append(np.array(c, copy=True))
        return real_linprog(c, *a, **k)

    ref.linprog = spy
    rng = np.random.default_rng(77)
    out: dict[str, np.ndarray] = {}
    cases = [(12, 64, 0, 400), (40, 128, 7, 600), (150, 256, 0, 2048), (90, 1024, 11, 1500), (5, 32, 0, 10_000)]
    meta = []
    for ci, (n, dim, heading_every, max_size) in enumerate(cases):
        chunklets, X = make_document(rng, n, dim, heading_every)
        captured.clear()
        chunks, chunk_embeddings = ref.split_chunks(chunklets, X, max_size=max_size)
        sizes = [len(m) for m in chunk_embeddings]
        out[f"case{ci}_X"] = X
        out[f"case{ci}_sizes"] = np.asarray(sizes, dtype=np.int64)
        out[f"case{ci}_cost"] = captured[0] if captured else np.zeros(0, np.float32)
        meta.append({"chunklets": chunklets, "max_size": max_size, "chunks": chunks})
    out["meta_json"] = np.asarray(json.dumps(meta))
    np.savez(OUT, **out)
    print(f"wrote {OUT}: " + ", ".join(f"case{i}: {len(out[f'case{i}_sizes'])} chunks" for i in range(len(cases))))


if __name__ == "__main__":
    main()
