"""raglite_amd -- MI355X-native retrieval / rerank hot path for RAGLite.

Hand-written HIP (gfx950) kernels behind a C ABI (`include/raglite_hip.h`, `libraglite_hip.so`)
and a Python host layer that mirrors the reference's own entry points for this path:
`embed_strings()`, `vector_search()`, `rerank_chunks()` and the `RAGLiteConfig.search_method` /
`.reranker` plugin objects.  There is no CPU fallback: without the shared library (or a gfx950
GPU) the calls raise.
"""

from raglite_amd._chunking import partition_cost, partition_similarities, split_chunks
from raglite_amd._config import HotPathConfig
from raglite_amd._embed import (
    embed_strings,
    embed_strings_with_late_chunking,
    embed_strings_without_late_chunking,
    embedding_type,
    set_embedder_factory,
)
from raglite_amd._ops import (
    DeviceIndex,
    adapter_apply,
    merge_topk,
    pack_bits,
    get_default_option,
    pool_norm,
    set_default_option,
    set_device,
    synth_fill,
    topk,
)
from raglite_amd._search import (
    GpuIndex,
    hybrid_search,
    reciprocal_rank_fusion,
    GpuVectorSearch,
    MaxSimRanker,
    attach_index,
    detach_index,
    rerank_chunks,
    search_and_rerank_chunks,
    select_reranker,
    set_language_detector,
    vector_search,
)
from raglite_amd._cross_encoder import CrossEncoderShape, TorchCrossEncoderRanker
from raglite_amd._torch_embedder import EncoderShape, HashTokenizer, SentencePieceTokenizer, TorchTokenEmbedder
from raglite_amd._query_adapter import update_query_adapter
from raglite_amd._comm import Communicator
from raglite_amd._sharded import ShardedIndex, merge_topk_host, shard_bounds_by_chunk

__all__ = [
    "SentencePieceTokenizer",
    "set_default_option",
    "get_default_option",
    "partition_cost",
    "partition_similarities",
    "split_chunks",
    "hybrid_search",
    "reciprocal_rank_fusion",
    "update_query_adapter",
    "EncoderShape",
    "HashTokenizer",
    "TorchTokenEmbedder",
    "TorchCrossEncoderRanker",
    "CrossEncoderShape",
    "pack_bits",
    "Communicator", "DeviceIndex", "GpuIndex", "GpuVectorSearch", "HotPathConfig", "MaxSimRanker", "ShardedIndex",
    "adapter_apply", "attach_index", "detach_index", "embed_strings", "embed_strings_with_late_chunking",
    "embed_strings_without_late_chunking", "embedding_type", "merge_topk", "merge_topk_host", "pool_norm",
    "rerank_chunks", "search_and_rerank_chunks", "select_reranker", "set_language_detector", "set_device", "set_embedder_factory", "shard_bounds_by_chunk",
    "synth_fill", "topk", "vector_search",
]
__version__ = "0.1.0"
