"""The subset of `RAGLiteConfig` the hot path reads.

The functions in this package accept the reference's real `raglite.RAGLiteConfig`
(`src/raglite/_config.py:42-83`) -- they only read attributes -- or this stand-in with the same field
names and defaults for environments where RAGLite itself is not installed (the GPU box).
Fields of the reference config that the path never touches (db_url, llm, ...) are omitted.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Literal

DEFAULT_CHUNK_MAX_SIZE = 2048  # RAGLiteConfig.chunk_max_size default, `_config.py:67`, used by `_search.py:66`


@dataclass(frozen=True)
class HotPathConfig:
    embedder: str = "llama-cpp-python/lm-kit/bge-m3-gguf/*F16.gguf@512"
    embedder_normalize: bool = True
    chunk_max_size: int = DEFAULT_CHUNK_MAX_SIZE
    vector_search_distance_metric: Literal["cosine", "dot", "l2"] = "cosine"
    vector_search_multivector: bool = True
    vector_search_query_adapter: bool = True
    reranker: Any = field(default=None, compare=False)
    search_method: Any = field(default=None, compare=False)
    self_query: bool = False
