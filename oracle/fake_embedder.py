"""Deterministic stand-in for `llama_cpp.Llama(embedding=True, pooling_type=NONE)` -- TEST INFRASTRUCTURE.

The reference obtains token-level embeddings from llama.cpp (`src/raglite/_embed.py:64-66,119,
151-154`), which is not installable here.  This fake exposes exactly the duck-typed surface
`_embed.py` touches -- `n_ctx()`, `n_batch`, `tokenize(bytes, add_bos=)`, `detokenize(list[int])`,
`embed(str | list[str])` -- with a deterministic tokenizer and deterministic fp32 token
embeddings (returned as Python floats, like llama-cpp-python does), so that

* `oracle/make_golden.py` can run the REAL reference `_embed.py` on it and record outputs, and
* the tests can feed the same token embeddings to the oracle and to the HIP path.

The embedder adds two special rows (BOS/EOS) to every `embed()` call so that the reference's
largest-remainder apportioning (`_embed.py:122-129`) is exercised with a non-zero remainder.
"""

from __future__ import annotations

import numpy as np

from oracle.oracle import synth_uniform

SENTINEL = "⊕"  # the reference's sentinel character, `_embed.py:70`


class FakeLlama:
    def __init__(self, dim: int = 64, n_ctx: int = 512, n_batch: int | None = None, seed: int = 7,
                 special_rows: int = 2) -> None:
        self.dim = dim
        self._n_ctx = n_ctx
        self.n_batch = n_ctx if n_batch is None else n_batch
        self.seed = seed
        self.special_rows = special_rows
        self._pieces: dict[int, str] = {}
        self.embed_calls = 0

    # -- tokenizer ---------------------------------------------------------------------
    def n_ctx(self) -> int:
        return self._n_ctx

    @staticmethod
    def _split(text: str) -> list[str]:
        pieces: list[str] = []
        i = 0
        while i < len(text):
            c = text[i]
            if c.isalnum():
                j = i
                while j < len(text) and j - i < 3 and text[j].isalnum():
                    j += 1
                pieces.append(text[i:j])
                i = j
            elif c == " " and i + 1 < len(text) and text[i + 1] == SENTINEL:
                pieces.append(" " + SENTINEL)  # a second sentinel variant, like real BPE vocabularies
                i += 2
            else:
                pieces.append(c)
                i += 1
        return pieces

    def _token_id(self, piece: str) -> int:
        if piece == SENTINEL:
            tid = 999
        elif piece == " " + SENTINEL:
            tid = 998
        else:
            h = 2166136261
            for b in piece.encode():
                h = ((h ^ b) * 16777619) & 0xFFFFFFFF
            tid = 1000 + h % 30000
            # Resolve hash collisions deterministically so detokenize stays a function.
            while tid in self._pieces and self._pieces[tid] != piece:
                tid += 1
        self._pieces[tid] = piece
        return tid

    def tokenize(self, data: bytes, add_bos: bool = False) -> list[int]:  # noqa: FBT001,FBT002
        toks = [self._token_id(p) for p in self._split(data.decode())]
        return ([1] if add_bos else []) + toks

    def detokenize(self, tokens: list[int]) -> bytes:
        return "".join(self._pieces.get(t, "") for t in tokens).encode()

    # -- embeddings --------------------------------------------------------------------
    def token_matrix(self, text: str) -> np.ndarray:
        """(T, dim) float32 token embeddings for `text`; T = #tokens + special rows."""
        toks = self.tokenize(text.encode(), add_bos=False)
        assert len(toks) + self.special_rows <= self._n_ctx, "segment exceeds fake n_ctx"
        ids = ([1] if self.special_rows >= 1 else []) + toks + ([2] if self.special_rows >= 2 else [])
        rows = [
            synth_uniform(self.seed * 1_000_003 + t * 131 + 7, r * self.dim, self.dim)
            + np.float32(0.25) * synth_uniform(self.seed + 17, (r % 64) * self.dim, self.dim)
            for r, t in enumerate(ids)
        ]
        return np.vstack(rows).astype(np.float32)

    def embed(self, text):  # noqa: ANN001,ANN201 - mirrors llama_cpp.Llama.embed
        self.embed_calls += 1
        if isinstance(text, str):
            return self.token_matrix(text).tolist()
        return [self.token_matrix(t).tolist() for t in text]


def make_sentences(seed: int, n: int, min_words: int = 3, max_words: int = 24) -> list[str]:
    """Deterministic pseudo-prose; every sentence ends with '. ' like split_sentences output."""
    bits = (synth_uniform(seed, 0, n * (max_words + 1) * 2) + 1.0) * 0.5
    out, p = [], 0
    for _ in range(n):
        nw = min_words + int(bits[p] * (max_words - min_words + 1)); p += 1
        words = []
        for _w in range(nw):
            ln = 1 + int(bits[p] * 9); p += 1
            base = int(bits[p] * 1e6); p += 1
            words.append("".join(chr(ord("a") + (base // (26**k)) % 26) for k in range(ln)))
        out.append(" ".join(words).capitalize() + ". ")
    return out
