"""Host mirror of the retrieval half of `src/raglite/_search.py` over a device-resident index.

    vector_search(query, *, num_results=3, oversample=4, metadata_filter=None, config=None)
        -> (list[ChunkId], list[float])                                   (`_search.py:36-153`)
    rerank_chunks(query, chunk_ids, *, config=None) -> list[chunk]        (`_search.py:364-397`)
    search_and_rerank_chunks(...)                                         (`_search.py:400-414`)
    GpuVectorSearch   -- a `BasicSearchMethod` (`_typing.py:35-43`) for `RAGLiteConfig.search_method`
    MaxSimRanker      -- a duck-typed `rerankers.BaseRanker` for `RAGLiteConfig.reranker`
                         (`.rank(query=, docs=)` -> `.results[*].doc_id`, `_search.py:394-396`)

The reference evaluates distance + ORDER BY/LIMIT + GROUP BY inside DuckDB/pgvector; here the
`chunk_embedding` table lives in HBM (`GpuIndex`) and the same three steps run as HIP kernels.  The
store, ORM and HNSW index are out of scope.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

import numpy as np

from raglite_amd import _ops
from raglite_amd._config import DEFAULT_CHUNK_MAX_SIZE, HotPathConfig
from raglite_amd._embed import embed_strings

ChunkId = str


class GpuIndex:
    """Device image of the `chunk` / `chunk_embedding` tables for one database.

    chunk_ids        list[str] -- `Chunk.id` per chunk ordinal (`_database.py:233`)
    chunk_embeddings list of (n_i, dim) matrices -- `Chunk.embedding_matrix` (`_database.py:279-283`),
                     or a single (N, dim) matrix together with `chunk_offsets`
    query_adapter    optional (dim, dim) matrix -- `IndexMetadata.get("default")["query_adapter"]`
                     (`_search.py:58-62`, fitted by `_query_adapter.py:141-219`, out of scope)
    docs             optional list[str] -- `str(chunk)` per chunk, lets `MaxSimRanker` map the strings the
                     reranker plugin receives back to chunk ordinals
    metadata         optional list[dict] per chunk for `metadata_filter`
    storage          "f32", or "f16" (the reference's own storage precision)
    exact_fp32       multiply with exact fp32 MFMAs instead of the default fp16 (hi, lo) split of fp32 operands
    """

    def __init__(self, chunk_ids: Sequence[ChunkId], chunk_embeddings, *, chunk_offsets=None,
                 metric: str = "cosine", query_adapter=None, docs: Sequence[str] | None = None,
                 metadata: Sequence[dict] | None = None, storage: str = "f32", exact_fp32: bool = False) -> None:
        if chunk_offsets is None:
            mats = [np.asarray(m, dtype=np.float32).reshape(len(m), -1) for m in chunk_embeddings]
            sizes = np.asarray([len(m) for m in mats], dtype=np.int64)
            chunk_offsets = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
            dim = mats[0].shape[1] if mats else 0
            matrix = np.vstack(mats) if mats else np.zeros((0, max(dim, 1)), dtype=np.float32)
        else:
            matrix = chunk_embeddings
        if len(chunk_ids) != len(chunk_offsets) - 1:
            raise ValueError("one chunk id per chunk is required")
        self.chunk_ids = list(chunk_ids)
        self.index = _ops.DeviceIndex(matrix, chunk_offsets, metric=metric, storage=storage)
        if exact_fp32:  # ordered fp32 MFMA chain instead of the fp16 (hi, lo) split (include/raglite_hip.h, rl_index_set_arithmetic)
            self.index.set_exact_fp32()
        self.metric = metric
        self.query_adapter = None if query_adapter is None else np.asarray(query_adapter, dtype=np.float32)
        self.docs = None if docs is None else list(docs)
        self._doc_to_ordinal = None if docs is None else {d: i for i, d in enumerate(self.docs)}
        self.metadata = None if metadata is None else list(metadata)
        self._id_to_ordinal = {cid: i for i, cid in enumerate(self.chunk_ids)}

    def ordinal_of(self, chunk_id: ChunkId) -> int:
        return self._id_to_ordinal[chunk_id]

    def ordinal_of_doc(self, doc: str) -> int:
        if self._doc_to_ordinal is None:
            raise ValueError("GpuIndex was built without `docs`; MaxSimRanker cannot map strings to chunks")
        return self._doc_to_ordinal[doc]

    # -- lifecycle -------------------------------------------------------------------------
    def insert_chunks(self, chunk_ids: Sequence[ChunkId], chunk_embeddings, *, docs: Sequence[str] | None = None,
                      metadata: Sequence[dict] | None = None) -> None:
        """`insert_documents` on the device image (`src/raglite/_insert.py:247-272`): append the chunks'
        embedding matrices; existing ordinals keep their meaning."""
        mats = [np.asarray(m, dtype=np.float32).reshape(len(m), -1) for m in chunk_embeddings]
        if len(mats) != len(chunk_ids):
            raise ValueError("one embedding matrix per chunk id is required")
        if any(cid in self._id_to_ordinal for cid in chunk_ids):
            raise ValueError("chunk id already present")  # the reference skips existing documents (`_insert.py:184-186`)
        if (self.docs is None) != (docs is None) or (self.metadata is None) != (metadata is None):
            raise ValueError("docs / metadata must be given iff the index was built with them")
        if not mats:
            return
        self.index.append(np.vstack(mats), np.asarray([len(m) for m in mats], dtype=np.int64))
        base = len(self.chunk_ids)
        self.chunk_ids.extend(chunk_ids)
        self._id_to_ordinal.update({cid: base + i for i, cid in enumerate(chunk_ids)})
        if docs is not None:
            self.docs.extend(docs)
            self._doc_to_ordinal.update({d: base + i for i, d in enumerate(docs)})
        if metadata is not None:
            self.metadata.extend(metadata)

    def delete_chunks(self, chunk_ids: Sequence[ChunkId]) -> int:
        """`delete_documents` on the device image (`src/raglite/_delete.py:148-176`): the chunks never match
        again; unknown ids are ignored like the reference's `WHERE id IN (...)`.  Returns the number deleted."""
        ords = [self._id_to_ordinal.pop(cid) for cid in chunk_ids if cid in self._id_to_ordinal]
        if ords:
            self.index.delete_chunks(np.asarray(ords, dtype=np.int64))
        return len(ords)

    # -- the real store ----------------------------------------------------------------------
    @classmethod
    def from_store(cls, bind: Any, *, metric: str = "cosine", storage: str = "f32", exact_fp32: bool = False) -> "GpuIndex":
        """Build the device index from a RAGLite database: `chunk_embedding` rows ordered by (chunk_id, id)
        (`src/raglite/_database.py:403-430`), the chunks' `str(chunk)` text and metadata, and the stored query adapter
        (`:450-462`).  `bind`: SQLAlchemy Engine / Connection / Session or a database URL.  `metric` is the store's
        `vector_search_distance_metric` (`_config.py:69`).  The index remembers `bind` for `sync()`."""
        from raglite_amd import _store

        conn, owned = _store._connection(bind)  # noqa: SLF001
        try:
            img = _store.read_chunks(conn)
            adapter = _store.read_query_adapter(conn)
        finally:
            if owned:
                conn.close()
        if not img.rows:
            raise ValueError("First run `insert_documents()` to insert documents.")  # the reference's wording for an empty store
        off = np.concatenate(([0], np.cumsum(np.asarray(img.sizes, dtype=np.int64)))).astype(np.int64)
        gi = cls(img.chunk_ids, img.matrix(), chunk_offsets=off, metric=metric, query_adapter=adapter, docs=img.docs,
                 metadata=img.metadata, storage=storage, exact_fp32=exact_fp32)
        gi._bind = bind  # noqa: SLF001
        return gi

    def sync(self, bind: Any = None, *, compact_above: float = 0.25) -> tuple[int, int]:
        """Bring the device image up to date with the store after `insert_documents` / `delete_documents`
        (`src/raglite/_insert.py:247-272`, `src/raglite/_delete.py:148-176`): chunk ids that appeared are appended
        (rows ordered by `id` within a chunk), ids that vanished are tombstoned, the query adapter is re-read; when
        more than `compact_above` of the rows are dead the index is compacted.  Returns (appended, deleted)."""
        from raglite_amd import _store

        bind = bind if bind is not None else getattr(self, "_bind", None)
        if bind is None:
            raise ValueError("sync() needs the store: build the index with GpuIndex.from_store() or pass `bind`")
        conn, owned = _store._connection(bind)  # noqa: SLF001
        try:
            in_store = set(_store.list_embedded_chunk_ids(conn))
            gone = [cid for cid in self._id_to_ordinal if cid not in in_store]
            new = sorted(in_store.difference(self._id_to_ordinal))
            img = _store.read_chunks(conn, new) if new else None
            self.query_adapter = _store.read_query_adapter(conn)
        finally:
            if owned:
                conn.close()
        n_deleted = self.delete_chunks(gone)
        if img is not None and img.rows:
            mats, at = [], 0
            for size in img.sizes:
                mats.append(np.vstack(img.rows[at : at + size]))
                at += size
            self.insert_chunks(img.chunk_ids, mats, docs=img.docs if self.docs is not None else None,
                               metadata=img.metadata if self.metadata is not None else None)
        live_rows, _ = self.index.live()
        if self.index.n_rows and 1.0 - live_rows / self.index.n_rows > compact_above:
            self.compact()
        return (len(img.chunk_ids) if img is not None else 0), n_deleted

    def compact(self) -> None:
        """Drop the tombstoned chunks for good (`rl_index_compact`) and renumber the host-side tables accordingly."""
        remap = self.index.compact()
        keep = np.nonzero(remap >= 0)[0]
        if len(keep) == len(remap):
            return
        self.chunk_ids = [self.chunk_ids[i] for i in keep]
        self._id_to_ordinal = {cid: i for i, cid in enumerate(self.chunk_ids)}
        if self.docs is not None:
            self.docs = [self.docs[i] for i in keep]
            self._doc_to_ordinal = {d: i for i, d in enumerate(self.docs)}
        if self.metadata is not None:
            self.metadata = [self.metadata[i] for i in keep]

    def close(self) -> None:
        self.index.close()


# config (hashable, like the reference's lru_cache keys) -> GpuIndex
_attached: dict[Any, GpuIndex] = {}
_DEFAULT_KEY = "default"


def attach_index(index: GpuIndex, config: Any | None = None) -> None:
    """Make `index` the one `vector_search(..., config=config)` searches."""
    _attached[_DEFAULT_KEY if config is None else config] = index


def detach_index(config: Any | None = None) -> None:
    _attached.pop(_DEFAULT_KEY if config is None else config, None)


def _index_for(config: Any | None) -> GpuIndex:
    idx = _attached.get(_DEFAULT_KEY if config is None else config) or _attached.get(_DEFAULT_KEY)
    if idx is None:
        raise ValueError("No GpuIndex attached: call raglite_amd.attach_index(index, config) first.")
    return idx


def _adapt_metadata(metadata_filter: dict | None) -> dict | None:
    """Normalise filter values to lists (`src/raglite/_database.py` `_adapt_metadata`)."""
    if not metadata_filter:
        return None
    return {k: (list(v) if isinstance(v, (list, tuple, set)) else [v]) for k, v in metadata_filter.items()}


def _matches(meta: dict, flt: dict) -> bool:
    """JSON containment `metadata @> filter` (`_search.py:84-97`): every filter value must occur."""
    for key, wanted in flt.items():
        have = meta.get(key)
        have = list(have) if isinstance(have, (list, tuple, set)) else [have]
        if any(w not in have for w in wanted):
            return False
    return True


def vector_search(query: str | np.ndarray, *, num_results: int = 3, oversample: int = 4,
                  metadata_filter: dict | None = None, config: Any | None = None,
                  index: GpuIndex | None = None) -> tuple[list[ChunkId], list[float]]:
    """Search chunks with an exact GPU scan (the reference's HNSW search is approximate)."""
    cfg = config or HotPathConfig()
    gi = index or _index_for(config)
    metadata_filter = _adapt_metadata(metadata_filter)
    if getattr(cfg, "self_query", False) and isinstance(query, str):
        raise NotImplementedError("self_query needs the LLM stack, which is outside this package")
    # Embed the query (`_search.py:54-56`).
    q = embed_strings([query], config=cfg)[0, :] if isinstance(query, str) else np.ravel(query)
    # Apply the query adapter (`_search.py:58-62`): result is cast back to the query dtype.
    if cfg.vector_search_query_adapter and gi.query_adapter is not None:
        q = _ops.adapter_apply(gi.query_adapter, np.asarray(q, dtype=np.float32)).astype(q.dtype)
    if gi.index.n_rows == 0:
        return [], []  # empty database (`tests/test_search.py:76-85`)
    # `_search.py:66-67`
    corrected_oversample = oversample * cfg.chunk_max_size / DEFAULT_CHUNK_MAX_SIZE
    num_hits = round(corrected_oversample) * max(num_results, 10)
    if num_hits < 1 or num_results < 1:
        return [], []
    if metadata_filter:
        return _filtered_search(gi, q, num_hits, num_results, metadata_filter)
    _check_limits(num_hits, num_results)
    scores, chunks, count = gi.index.search_chunks(np.asarray(q, dtype=np.float32), num_hits, num_results)
    n = int(count)
    return [gi.chunk_ids[c] for c in chunks[:n].tolist()], [float(s) for s in scores[:n]]


def _check_limits(num_hits: int, num_results: int) -> None:
    """The exact selection ranks at most K_MAX rows per query; asking for more is an error, not a silent truncation
    (the reference's `LIMIT num_hits` has no such bound: a documented limit of this implementation)."""
    if num_hits > _ops.K_MAX or num_results > _ops.K_MAX:
        raise ValueError(f"vector_search: num_results={num_results} needs the top {num_hits} rows, more than the "
                         f"{_ops.K_MAX} the exact top-k kernel ranks; lower num_results or oversample")


FILTER_FIRST_MAX_ROWS = 100_000  # `metadata_count <= 100_000` (`src/raglite/_search.py:105`)
ORDER_FIRST_LIMIT = 1_000_000    # `.limit(1_000_000)` (`src/raglite/_search.py:124`)


def _filtered_search(gi: GpuIndex, q, num_hits: int, num_results: int, flt: dict):
    """The reference's filtered vector search (`_search.py:96-141`): evaluate the JSON containment on the host metadata and
    push the result down as a bitset over chunk ordinals.  Like the reference, count the matching embedding rows first
    (`:97-103`): up to 100 000 -> filter first, rank the matching rows (`:105-119`); more -> order first, i.e. only the
    1 000 000 rows nearest to the query are eligible, then the filter (`:120-141`).  Both branches rank exactly; they
    differ only on a corpus of more than 1 000 000 rows."""
    if gi.metadata is None:
        raise ValueError("GpuIndex was built without `metadata`; metadata_filter cannot be applied")
    allowed = np.fromiter((_matches(m, flt) for m in gi.metadata), dtype=bool, count=len(gi.metadata))
    if not allowed.any():
        return [], []
    _check_limits(num_hits, num_results)
    offsets = gi.index.chunk_offsets
    rows_per_chunk = np.diff(offsets) if offsets is not None else np.ones(len(allowed), dtype=np.int64)
    matching_rows = int(rows_per_chunk[allowed].sum())
    rank_limit = ORDER_FIRST_LIMIT if matching_rows > FILTER_FIRST_MAX_ROWS else None
    scores, chunks, count = gi.index.search_chunks(np.asarray(q, dtype=np.float32), num_hits, num_results,
                                                   chunk_filter=allowed, rank_limit=rank_limit)
    n = int(count)
    return [gi.chunk_ids[c] for c in chunks[:n].tolist()], [float(s) for s in scores[:n]]


class GpuVectorSearch:
    """`BasicSearchMethod` for `RAGLiteConfig.search_method` (`src/raglite/_typing.py:35-43`,
    consumed at `src/raglite/_rag.py:53-63`)."""

    def __init__(self, index: GpuIndex, *, oversample: int = 4) -> None:
        self.index = index
        self.oversample = oversample

    def __call__(self, query: str | np.ndarray, *, num_results: int = 8, metadata_filter: dict | None = None,
                 config: Any | None = None) -> tuple[list[ChunkId], list[float]]:
        return vector_search(query, num_results=num_results, oversample=self.oversample,
                             metadata_filter=metadata_filter, config=config, index=self.index)


# ---- hybrid search ------------------------------------------------------------------------
def reciprocal_rank_fusion(rankings: Sequence[Sequence[ChunkId]], *, k: int = 60,
                           weights: Sequence[float] | None = None) -> tuple[list[ChunkId], list[float]]:
    """Reciprocal Rank Fusion (`src/raglite/_search.py:233-252`): score(id) = sum_r w_r / (k + rank_r(id)), ranked by
    descending score; ids with equal scores keep the order in which they were first seen (stable sort)."""
    if weights is None:
        weights = [1.0] * len(rankings)
    if len(weights) != len(rankings):
        raise ValueError("The number of weights must match the number of rankings.")
    score: dict[ChunkId, float] = {}
    for ranking, weight in zip(rankings, weights):
        for i, cid in enumerate(ranking):
            score[cid] = score.get(cid, 0.0) + weight / (k + i)
    if not score:
        return [], []
    ordered = sorted(score.items(), key=lambda kv: kv[1], reverse=True)
    return [cid for cid, _ in ordered], [s for _, s in ordered]


def hybrid_search(query: str | np.ndarray, *, num_results: int = 3, oversample: int = 2,
                  vector_search_weight: float = 0.75, keyword_search_weight: float = 0.25,
                  metadata_filter: dict | None = None, config: Any | None = None, index: GpuIndex | None = None,
                  keyword_search: Callable[..., tuple[list[ChunkId], list[float]]] | None = None,
                  ) -> tuple[list[ChunkId], list[float]]:
    """`src/raglite/_search.py:255-279`: GPU vector search fused with a keyword ranking by RRF.  The BM25 keyword
    search lives in the store (`_search.py:156-230`, out of scope): pass the reference's own `keyword_search` (or any
    callable with its signature) as `keyword_search=`; without one the fusion degenerates to the vector ranking."""
    vs_ids, _ = vector_search(query, num_results=oversample * num_results, metadata_filter=metadata_filter,
                              config=config, index=index)
    ks_ids: list[ChunkId] = []
    if keyword_search is not None:
        ks_ids, _ = keyword_search(query, num_results=oversample * num_results, metadata_filter=metadata_filter,
                                   config=config)
    ids, scores = reciprocal_rank_fusion([vs_ids, ks_ids], weights=[vector_search_weight, keyword_search_weight])
    return ids[:num_results], scores[:num_results]


# ---- reranking -------------------------------------------------------------------------------------------
@dataclass
class Result:
    """Shape of `rerankers.results.Result` that `_search.py:396` reads (`.doc_id`), plus score and rank."""

    doc_id: int
    score: float
    rank: int
    text: str = ""


@dataclass
class RankedResults:
    results: list[Result] = field(default_factory=list)
    query: str = ""

    def top_k(self, k: int) -> list[Result]:
        return self.results[:k]


class MaxSimRanker:
    """ColBERT-style late-interaction reranker behind the reference's reranker plugin boundary.

    `rank(query=, docs=)` scores every doc as  sum_i max_j q_i . d_j  over the query's token vectors
    q_i and the doc's chunklet vectors d_j (GPU: `rl_maxsim_rerank`) and returns them best-first.
    `query_encoder(query: str) -> (nq, dim)` supplies the query's multi-vector representation.
    """

    def __init__(self, index: GpuIndex, query_encoder: Callable[[str], np.ndarray]) -> None:
        self.index = index
        self.query_encoder = query_encoder

    @classmethod
    def from_embedder(cls, index: GpuIndex, embedder: Any, *, normalize: bool = True, max_vectors: int = 32) -> "MaxSimRanker":
        """Query side from a token-level embedder (`raglite_amd.TorchTokenEmbedder`, or llama.cpp with pooling NONE):
        the query's token embeddings, L2-normalised like the stored chunklet vectors, at most `max_vectors` of them
        (ColBERT's fixed query length; the first tokens are kept)."""

        def encode(query: str) -> np.ndarray:
            m = embedder.embed(query)
            m = m.float().cpu().numpy() if hasattr(m, "cpu") else np.asarray(m, dtype=np.float32)
            m = m[:max_vectors]
            if normalize:
                m = m / np.maximum(np.linalg.norm(m, axis=1, keepdims=True), np.finfo(np.float32).eps)
            return m.astype(np.float32)

        return cls(index, encode)

    def score(self, query: str, ordinals: Sequence[int]) -> np.ndarray:
        qv = np.asarray(self.query_encoder(query), dtype=np.float32)
        qv = qv.reshape(1, *qv.shape) if qv.ndim == 2 else qv.reshape(1, 1, -1)
        cand = np.asarray(ordinals, dtype=np.int32).reshape(1, -1)
        return np.asarray(self.index.index.maxsim_rerank(qv, cand))[0]

    def rank(self, query: str, docs: Sequence[Any], doc_ids: Sequence[int] | None = None, **_: Any) -> RankedResults:
        docs = list(docs)
        if not docs:
            return RankedResults([], query)
        ordinals = [self.index.ordinal_of_doc(d if isinstance(d, str) else str(d)) for d in docs]
        scores = self.score(query, ordinals)
        # best first; equal scores keep input order (stable), NaN last
        key = np.where(np.isnan(scores), -np.inf, scores)
        order = np.lexsort((np.arange(len(docs)), -key))
        ids = list(doc_ids) if doc_ids is not None else list(range(len(docs)))
        return RankedResults(
            [Result(doc_id=ids[i], score=float(scores[i]), rank=r + 1, text=str(docs[i])) for r, i in enumerate(order)],
            query,
        )


_language_detector: Callable[[str], str] | None = None
_language_detector_set = False


def set_language_detector(detect: Callable[[str], str] | None) -> None:
    """Install the callable `rerank_chunks` uses to pick a language-specific reranker from a `dict` of rerankers
    (`src/raglite/_search.py:379-392`).  The reference imports `langdetect.detect`; any `str -> language code` callable
    works (fastText, lingua, a customer's own).  None restores the default (langdetect if installed, else no detection)."""
    global _language_detector, _language_detector_set
    _language_detector, _language_detector_set = detect, detect is not None


def _default_language_detector() -> Callable[[str], str] | None:
    if _language_detector_set:
        return _language_detector
    try:
        from langdetect import detect  # the reference's dependency (`_search.py:13`); not in this image

        return detect
    except ImportError:
        return None


def select_reranker(reranker: Any, query: str, chunks: Sequence[Any], detect: Callable[[str], str] | None = None) -> Any:
    """The reference's reranker selection (`src/raglite/_search.py:378-392`): a `dict` maps language codes (and "other")
    to rankers; when every chunk and the query are detected as ONE language that has an entry, that ranker is used,
    otherwise the "other" entry.  A failing detector (the reference suppresses `LangDetectException`) or no detector at
    all falls back to "other"."""
    if not isinstance(reranker, dict):
        return reranker
    detect = detect or _default_language_detector()
    langs: set[str] = set()
    if detect is not None:
        try:
            langs = {detect(str(chunk)) for chunk in chunks}
            langs.add(detect(query))
        except Exception:  # noqa: BLE001 - e.g. langdetect's LangDetectException on text without letters
            langs = set()
    if len(langs) == 1 and (lang := next(iter(langs))) in reranker:
        return reranker[lang]
    return reranker.get("other")


def rerank_chunks(query: str, chunk_ids: Sequence[Any], *, config: Any | None = None,
                  chunk_lookup: Callable[[Sequence[ChunkId]], list[Any]] | None = None,
                  detect: Callable[[str], str] | None = None) -> list[Any]:
    """Rerank chunks according to their relevance to a query (`src/raglite/_search.py:364-397`).

    `chunk_ids` may be chunk ids or chunk objects (anything whose `str()` is the chunk text).  Ids are
    resolved through `chunk_lookup` (the reference uses `retrieve_chunks`, a SQL query, out of scope).
    `detect`: language detector for a `dict` of rerankers (default: `set_language_detector` / langdetect)."""
    cfg = config or HotPathConfig()
    chunks = list(chunk_ids)
    if chunks and all(isinstance(c, ChunkId) for c in chunks):
        if chunk_lookup is None:
            raise ValueError("chunk ids need a chunk_lookup callable (the reference's retrieve_chunks)")
        chunks = chunk_lookup(chunks)
    reranker = getattr(cfg, "reranker", None)
    if not reranker or not chunks:
        return chunks
    reranker = select_reranker(reranker, query, chunks, detect)
    if reranker:
        results = reranker.rank(query=query, docs=[str(chunk) for chunk in chunks])
        chunks = [chunks[result.doc_id] for result in results.results]
    return chunks


def search_and_rerank_chunks(query: str, *, num_results: int = 8, oversample: int = 4,
                             search: Callable[..., tuple[list[ChunkId], list[float]]] = vector_search,
                             config: Any | None = None, metadata_filter: dict | None = None,
                             chunk_lookup: Callable[[Sequence[ChunkId]], list[Any]] | None = None) -> list[Any]:
    """`src/raglite/_search.py:400-414` (default `search` there is hybrid_search, whose keyword half is SQL)."""
    chunk_ids, _ = search(query, num_results=oversample * num_results, metadata_filter=metadata_filter, config=config)
    return rerank_chunks(query, chunk_ids, config=config, chunk_lookup=chunk_lookup)[:num_results]
