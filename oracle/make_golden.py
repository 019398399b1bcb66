"""Generate `tests/golden/*.npz` by executing the REFERENCE's own `_embed.py` -- TEST INFRASTRUCTURE.

Run in the authoring container only (needs /root/reference, which does not exist on the GPU box):

    python -m oracle.make_golden

What is real and what is stubbed
--------------------------------
* REAL: `/root/reference/src/raglite/_embed.py` (all of it: token counting, segmenting,
  largest-remainder split, per-sentence `np.mean`, normalise, fp16 cast, the batch path),
  `_typing.py`, `_lazy_llama.py`, `_config.py` (the real frozen `RAGLiteConfig` dataclass).
* STUBBED (absent third-party wheels, none of which touches the arithmetic under test):
  `litellm`, `rerankers`, `raglite._litellm` (its only role in `_embed.py` is
  `LlamaCppPythonLLM.llm(...)` returning the llama embedder -> returns `FakeLlama`).
  The package `__init__` is bypassed (it imports the whole application) by registering an
  empty `raglite` package whose `__path__` points at the reference sources.

The fixtures hold the inputs (sentences, fake-embedder parameters) and the reference's
outputs; `tests/test_oracle_golden.py` re-derives the outputs with `oracle/oracle.py` and with
the host mirror in `raglite_amd/`, and the `-m gpu` tests push the same token matrices
through the HIP kernel.
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

REFERENCE_SRC = Path("/root/reference/src")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _install_stubs(fake_factory):
    pkg = types.ModuleType("raglite")
    pkg.__path__ = [str(REFERENCE_SRC / "raglite")]
    sys.modules["raglite"] = pkg

    litellm = types.ModuleType("litellm")
    litellm.embedding = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("API embedder not used"))
    sys.modules["litellm"] = litellm

    for name in ("rerankers", "rerankers.models", "rerankers.models.flashrank_ranker",
                 "rerankers.models.ranker"):
        sys.modules[name] = types.ModuleType(name)

    class _Ranker:  # constructed by RAGLiteConfig's default_factory; never called here
        def __init__(self, *a, **k) -> None:
            pass

    sys.modules["rerankers.models.flashrank_ranker"].FlashRankRanker = _Ranker
    sys.modules["rerankers.models.ranker"].BaseRanker = _Ranker

    rl_litellm = types.ModuleType("raglite._litellm")

    class LlamaCppPythonLLM:
        @staticmethod
        def llm(model: str, **kwargs):  # noqa: ANN003,ANN205
            return fake_factory(model)

    rl_litellm.LlamaCppPythonLLM = LlamaCppPythonLLM
    sys.modules["raglite._litellm"] = rl_litellm


def main() -> None:
    from oracle.fake_embedder import FakeLlama, make_sentences

    fakes: dict[str, FakeLlama] = {}

    def factory(model: str) -> FakeLlama:
        return fakes[model]

    _install_stubs(factory)
    from raglite._config import RAGLiteConfig  # REAL reference module
    from raglite import _embed as ref_embed  # REAL reference module

    OUT.mkdir(parents=True, exist_ok=True)
    cases = [
        # name, n_sentences, dim, n_ctx, n_batch, normalize, sentence seed
        ("late_short", 14, 64, 512, 512, True, 11),
        ("late_multiseg", 160, 64, 256, 256, True, 12),
        ("late_multiseg_nonorm", 90, 48, 200, 256, False, 13),
        ("late_d1024", 40, 1024, 512, 512, True, 14),
        ("late_single_sentence", 1, 128, 512, 512, True, 15),
    ]
    manifest = {}
    for name, n, dim, n_ctx, n_batch, normalize, sseed in cases:
        model = f"llama-cpp-python/fake/{name}@{n_ctx}"
        fakes[model] = FakeLlama(dim=dim, n_ctx=n_ctx, n_batch=n_batch, seed=sseed)
        sentences = make_sentences(sseed, n)
        cfg = RAGLiteConfig(llm="unused", embedder=model, embedder_normalize=normalize)
        out = ref_embed.embed_strings(sentences, config=cfg)  # dispatches to late chunking (:196-197)
        assert out.dtype == np.float16 and out.shape == (n, dim)
        np.savez_compressed(OUT / f"{name}.npz", output=out)
        manifest[name] = dict(kind="late_chunking", n_sentences=n, dim=dim, n_ctx=n_ctx, n_batch=n_batch,
                              normalize=normalize, sentence_seed=sseed, embedder_seed=sseed,
                              embed_calls=fakes[model].embed_calls)
    # a3: the batch (non-late-chunking) pooling path, `_embed_string_batch` :144-165, called directly
    # (with a llama-cpp embedder `embed_strings` never reaches it).
    for name, n, dim, normalize, sseed in [("batch_pool", 30, 96, True, 21), ("batch_pool_nonorm", 9, 64, False, 22)]:
        model = f"llama-cpp-python/fake/{name}@512"
        fakes[model] = FakeLlama(dim=dim, n_ctx=512, seed=sseed)
        strings = make_sentences(sseed, n)
        cfg = RAGLiteConfig(llm="unused", embedder=model, embedder_normalize=normalize)
        out = ref_embed.embed_strings_without_late_chunking(strings, config=cfg)
        assert out.dtype == np.float16 and out.shape == (n, dim)
        np.savez_compressed(OUT / f"{name}.npz", output=out)
        manifest[name] = dict(kind="batch", n_sentences=n, dim=dim, n_ctx=512, n_batch=512, normalize=normalize,
                              sentence_seed=sseed, embedder_seed=sseed)
    (OUT / "manifest.json").write_text(json.dumps(manifest, indent=1, sort_keys=True) + "\n")
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
