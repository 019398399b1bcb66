"""CPU oracle for the RAGLite retrieval/rerank hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the arithmetic the reference
(superlinear-ai/raglite v1.0.0, mounted at /root/reference when authoring) performs on
the path pool -> L2-normalise -> fp16 cast -> query-adapter matvec -> distance ->
row top-k -> per-chunk max -> chunk top-k, plus the multi-query MaxSim generalisation and
the shard merge.  Every function cites the reference file:line it follows.

Who may import this: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg.  Nothing under `raglite_amd/` imports it; the product path has no CPU fallback.

Parity pinning status
---------------------
* pool / normalise / cast (rows a1-a3 of SURVEY.md section 8): PINNED.  The reference's own
  `_embed.py` is executed in the authoring container by `oracle/make_golden.py` (third-party
  imports stubbed, a deterministic fake llama embedder supplying token embeddings) and its
  outputs are committed under `tests/golden/`; `tests/test_oracle_golden.py` checks this
  module against them bit-for-bit.
* the adapter's producer (`_optimize_query_target`, the closed-form
  Procrustes block, positive / negative row selection of `_query_adapter.py`): PINNED.
  `oracle/make_golden_adapter.py` executes the reference's own source lines (scipy + numpy are
  available) and `tests/golden/query_adapter.npz` holds their outputs.
* semantic-chunking similarities and reciprocal rank fusion (section 8f-4): PINNED.  The REAL
  `_split_chunks.split_chunks` runs in `oracle/make_golden_chunks.py` (cost vector captured at its
  `linprog` call); `reciprocal_rank_fusion` is exec'd from its source text.
* adapter application (a5, `_search.py:57-62`) and the num_hits rule (`:66-67`): PINNED -- the statements
  are exec'd from vector_search's body by `oracle/make_golden_adapter.py`.
* distance / two-stage selection (rows a6-a8): PARITY UNPINNED.  The reference
  evaluates these inside DuckDB (`array_cosine_distance` + usearch HNSW, approximate) or
  pgvector; neither engine nor any golden vector is available.  The
  restatement follows the SQL the reference emits (`_search.py:66-79,143-149`) with exact
  (brute-force) ranking; ties, which SQL leaves unspecified, resolve to the lowest row id /
  lowest chunk ordinal.
* MaxSim with several query vectors (row a9): new functionality behind the reference's
  reranker plugin; defined here as the multi-query generalisation of `_search.py:143-149` /
  `_query_adapter.py:174` and reduces to it for nq == 1 (tested).
"""

from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------
# Deterministic synthetic data (shared bit-for-bit with raglite_amd/csrc/synth.hip)
# ----------------------------------------------------------------------------------------

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


def synth_bits(seed: int, start: int, count: int) -> np.ndarray:
    """64 hash bits for elements [start, start+count) of stream `seed`."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0xD1342543DE82EF95)) & _M64
        return _splitmix64(idx ^ _splitmix64(np.full(1, key, dtype=np.uint64))[0])


def synth_uniform(seed: int, start: int, count: int) -> np.ndarray:
    """float32 uniform in [-1, 1): the top 24 hash bits u -> u * 2^-23 - 1 (exact in fp32)."""
    u = (synth_bits(seed, start, count) >> np.uint64(40)).astype(np.float32)
    return u * np.float32(2.0**-23) - np.float32(1.0)


def synth_small_int(seed: int, start: int, count: int) -> np.ndarray:
    """float32 integers in {-3..3}: dot products / squared norms are exact in fp32 for d<=1024."""
    u = (synth_bits(seed, start, count) >> np.uint64(40)) % np.uint64(7)
    return u.astype(np.float32) - np.float32(3.0)


def synth_matrix(seed: int, n_rows: int, dim: int, kind: str = "uniform", row0: int = 0) -> np.ndarray:
    gen = {"uniform": synth_uniform, "small_int": synth_small_int}[kind]
    return gen(seed, row0 * dim, n_rows * dim).reshape(n_rows, dim)


# ----------------------------------------------------------------------------------------
# a1: late-chunking pool   (reference: src/raglite/_embed.py:119-136)
# ----------------------------------------------------------------------------------------


def largest_remainder_sizes(n_token_rows: int, segment_tokens: np.ndarray) -> np.ndarray:
    """Rows of a segment's token matrix apportioned to its sentences.

    Follows `_embed.py:122-129` operation for operation (same `np.floor`, same `np.argsort`
    on the fractional parts, same `[-remainder:]` slice), because the tie behaviour of the
    largest-remainder method is defined by that argsort call.
    """
    segment_tokens = np.asarray(segment_tokens)
    sentence_size_frac = n_token_rows * (segment_tokens / np.sum(segment_tokens))
    sentence_size = np.floor(sentence_size_frac).astype(np.intp)
    remainder = n_token_rows - np.sum(sentence_size)
    if remainder > 0:
        top_remainders = np.argsort(sentence_size_frac - sentence_size)[-remainder:]
        sentence_size[top_remainders] += 1
    return sentence_size


def create_segment(
    content_start_index: int, max_tokens_preamble: int, max_tokens_content: int, num_tokens: np.ndarray
) -> tuple[int, int]:
    """Segment bounds with a <=38.2 % preamble.  Follows `_embed.py:38-58`."""
    cumsum_backwards = np.cumsum(num_tokens[:content_start_index][::-1])
    offset_preamble = np.searchsorted(cumsum_backwards, max_tokens_preamble, side="right")
    segment_start_index = content_start_index - int(offset_preamble)
    max_tokens_content = max_tokens_content + (
        max_tokens_preamble - np.sum(num_tokens[segment_start_index:content_start_index])
    )
    cumsum_forwards = np.cumsum(num_tokens[content_start_index:])
    offset_segment = np.searchsorted(cumsum_forwards, max_tokens_content, side="right")
    segment_end_index = content_start_index + int(offset_segment)
    return segment_start_index, segment_end_index


def create_segments(num_tokens: np.ndarray, n_ctx: int, n_batch: int) -> list[tuple[int, int, int]]:
    """All (segment_start, content_start, segment_end) triples.  Follows `_embed.py:99-110`."""
    max_tokens = min(n_ctx, n_batch) - 16
    max_tokens_preamble = round(0.382 * max_tokens)
    max_tokens_content = max_tokens - max_tokens_preamble
    segments = []
    content_start_index = 0
    while content_start_index < len(num_tokens):
        s, e = create_segment(content_start_index, max_tokens_preamble, max_tokens_content, num_tokens)
        segments.append((s, content_start_index, e))
        content_start_index = e
    return segments


def pool_segment(
    segment_embedding: np.ndarray, segment_tokens: np.ndarray, n_preamble_sentences: int
) -> np.ndarray:
    """Mean-pool the token rows of each *content* sentence.  Follows `_embed.py:122-135`.

    `segment_embedding` is float64 in the reference (llama-cpp returns Python floats,
    `np.asarray` at `_embed.py:119`); the mean is `np.mean(axis=0)` in float64.
    """
    segment_embedding = np.asarray(segment_embedding, dtype=np.float64)
    sizes = largest_remainder_sizes(len(segment_embedding), segment_tokens)
    mats = np.split(segment_embedding, np.cumsum(sizes)[:-1])
    content = [np.mean(m, axis=0, keepdims=True) for m in mats[n_preamble_sentences:]]
    return np.vstack(content)


def pool_spans(tokens: np.ndarray, span_begin: np.ndarray, span_end: np.ndarray) -> np.ndarray:
    """Span form of a1 (what the HIP kernel receives): mean of rows [b, e) per span, float64.

    An empty span yields NaN exactly as `np.mean` of a (0, d) matrix does in the reference
    (`_embed.py:132`, reachable when a sentence is apportioned zero rows).
    """
    tokens = np.asarray(tokens, dtype=np.float64)
    out = np.empty((len(span_begin), tokens.shape[1]), dtype=np.float64)
    for i, (b, e) in enumerate(zip(span_begin, span_end)):
        with np.errstate(invalid="ignore", divide="ignore"):
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out[i] = np.mean(tokens[b:e], axis=0)
    return out


# ----------------------------------------------------------------------------------------
# a2: L2 normalise + fp16 cast   (reference: _embed.py:138-140 and :158-164)
# ----------------------------------------------------------------------------------------


def l2_normalize(x: np.ndarray, eps: float | None = None) -> np.ndarray:
    """`X /= ||X||` rowwise.  eps=None: unguarded divide (`_embed.py:139`);
    eps given: `X /= max(norm, eps)` (`_embed.py:160-163`, eps = finfo(dtype).eps there)."""
    x = np.array(x, copy=True)
    norm = np.linalg.norm(x, axis=1, keepdims=True)
    if eps is not None:
        norm = np.maximum(norm, eps)
    with np.errstate(invalid="ignore", divide="ignore"):
        x /= norm
    return x


def to_fp16(x: np.ndarray) -> np.ndarray:
    """`astype(np.float16)` (round-to-nearest-even).  `_embed.py:140,164`."""
    with np.errstate(over="ignore"):
        return np.asarray(x).astype(np.float16)


def pool_norm_cast(
    tokens: np.ndarray, span_begin: np.ndarray, span_end: np.ndarray, normalize: bool = True,
    eps: float | None = None,
) -> tuple[np.ndarray, np.ndarray]:
    """a1+a2 on spans: returns (float64 pooled[/normalised], fp16 cast)."""
    x = pool_spans(tokens, span_begin, span_end)
    if normalize:
        x = l2_normalize(x, eps)
    return x, to_fp16(x)


def embed_string_batch_pool(token_matrices: list[np.ndarray], normalize: bool = True) -> np.ndarray:
    """a3: mean over all tokens of each string, eps-guarded normalise, fp16.  `_embed.py:154-164`."""
    embeddings = np.asarray([np.mean(np.asarray(row, dtype=np.float64), axis=0) for row in token_matrices])
    if normalize:
        eps = np.finfo(embeddings.dtype).eps
        norm = np.linalg.norm(embeddings, axis=1, keepdims=True)
        embeddings /= np.maximum(norm, eps)
    return embeddings.astype(np.float16)


# ----------------------------------------------------------------------------------------
# a5: query adapter   (reference: src/raglite/_search.py:58-62)
# ----------------------------------------------------------------------------------------


def adapter_apply(A: np.ndarray, q: np.ndarray) -> np.ndarray:
    """`(Q @ q).astype(q.dtype)`; batched form applies it to every row of q."""
    q = np.asarray(q)
    if q.ndim == 1:
        return (A @ q).astype(q.dtype)
    return (q @ A.T).astype(q.dtype)


# ----------------------------------------------------------------------------------------
# a6: distance / similarity   (reference: _search.py:69-72, _typing.py:123-134)
# ----------------------------------------------------------------------------------------


def distance(E: np.ndarray, q: np.ndarray, metric: str = "cosine", dtype=np.float64) -> np.ndarray:
    """DuckDB semantics: cosine -> 1 - e.q/(|e||q|); dot -> -(e.q); l2 -> |e-q|_2.

    `dtype=np.float64` is the "truth" variant, `np.float32` the as-computed variant (DuckDB
    evaluates FLOAT[d] arrays in fp32; its exact rounding is unverifiable here).

    Cosine, the two published forms.  DuckDB (>= 1.1, the reference's pin `duckdb>=1.1.3`, absent from /root/reference
    and from this image) documents `array_cosine_distance(a, b)` as `1 - array_cosine_similarity(a, b)` with
    `array_cosine_similarity = sum(a_i b_i) / sqrt(sum(a_i^2) * sum(b_i^2))`: ONE square root of the product of the two
    squared norms.  pgvector's `<=>` computes `1 - dot / sqrt(norm_a * norm_b)` the same way.  This restatement and
    the device (scan.hip:finish_score, transform_kernel, the GEMM epilogue) divide by `sqrt(sum a^2) * sqrt(sum b^2)`:
    TWO roots and a product, because ||e|| is precomputed per row when the index is built.  In IEEE arithmetic the two
    forms differ by the roundings of one extra sqrt and one multiply: at most 1.5 ulp of the denominator, i.e. <= 2 ulp
    (2.4e-7 relative) of the similarity -- three orders of magnitude inside the 1e-4 bar, and it cannot change a ranking
    of scores that differ by more than that.  On integer-valued data with perfect-square norms both forms are exact.
    `distance_duckdb_fp32` below restates DuckDB's own formulation in float32 so that the gap is MEASURED, not argued."""
    E = np.asarray(E, dtype=dtype)
    q = np.asarray(q, dtype=dtype)
    if metric == "cosine":
        with np.errstate(invalid="ignore", divide="ignore"):
            return 1.0 - (E @ q) / (np.linalg.norm(E, axis=1) * np.linalg.norm(q))
    if metric == "dot":
        return -(E @ q)
    if metric == "l2":
        return np.linalg.norm(E - q[None, :], axis=1)
    raise ValueError(f"Unsupported metric: {metric}")


def _seq_dot_f32(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """sum_k A[:, k] * B[:, k] accumulated in ELEMENT ORDER in float32 (acc += x * y, one rounding per product and per add, as a scalar
    C++ loop over `float` without FMA contraction computes it)."""
    acc = np.zeros(A.shape[0], dtype=np.float32)
    for k in range(A.shape[1]):
        acc = (acc + (A[:, k] * B[:, k]).astype(np.float32)).astype(np.float32)
    return acc


def distance_duckdb_fp32(E: np.ndarray, q: np.ndarray, metric: str = "cosine") -> np.ndarray:
    """The as-computed variant for a6 in DuckDB's own formulation -- parity stays UNPINNED (DuckDB >= 1.1.3 is the reference's
    dependency, `pyproject.toml`, absent from /root/reference and from this image; what follows restates its published
    `array_cosine_distance` / `array_negative_inner_product` / `array_distance` for FLOAT[d], the functions the reference's SQL calls at
    `src/raglite/_typing.py:123-134`): everything in float32, one pass over the elements in order,
        cosine  1 - clamp(dot / sqrt(norm_a * norm_b), -1, 1)     ONE square root of the product of the squared norms, clamped
        dot     -(dot)                                            (array_negative_inner_product)
        l2      sqrt(sum (a_i - b_i)^2)
    DuckDB may vectorise the loop (another summation order, same bound); the element-order sum is the canonical scalar form.  This
    function exists to MEASURE how far the formulation used by this repository (`distance(..., np.float32)`: two roots and a product,
    pairwise / blocked sums) is from it -- tests/test_oracle_props.py bounds the gap in float32 ulps on fp16-rounded unit rows (what
    RAGLite stores, `_embed.py:138-140`), tests/test_gpu_parity.py holds the kernels to that bound + 2 ulp."""
    E = np.ascontiguousarray(E, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    Q = np.broadcast_to(q[None, :], E.shape)
    one = np.float32(1.0)
    if metric == "cosine":
        dot, na, nb = _seq_dot_f32(E, Q), _seq_dot_f32(E, E), _seq_dot_f32(Q, Q)
        with np.errstate(invalid="ignore", divide="ignore"):
            sim = (dot / np.sqrt((na * nb).astype(np.float32)).astype(np.float32)).astype(np.float32)
        sim = np.where(np.isnan(sim), sim, np.clip(sim, np.float32(-1.0), one)).astype(np.float32)
        return (one - sim).astype(np.float32)
    if metric == "dot":
        return (-_seq_dot_f32(E, Q)).astype(np.float32)
    if metric == "l2":
        D = (E - Q).astype(np.float32)
        return np.sqrt(_seq_dot_f32(D, D)).astype(np.float32)
    raise ValueError(f"Unsupported metric: {metric}")


def similarity_duckdb_fp32(E: np.ndarray, q: np.ndarray, metric: str = "cosine") -> np.ndarray:
    """`sim = 1.0 - dist` (`_search.py:72`) over `distance_duckdb_fp32`, in float32 (DuckDB FLOAT arithmetic)."""
    return (np.float32(1.0) - distance_duckdb_fp32(E, q, metric)).astype(np.float32)


def similarity(E: np.ndarray, q: np.ndarray, metric: str = "cosine", dtype=np.float64) -> np.ndarray:
    """`sim = 1.0 - dist` (`_search.py:72`)."""
    return 1.0 - distance(E, q, metric, dtype)


# ----------------------------------------------------------------------------------------
# a7 + a8: two-stage selection   (reference: _search.py:66-67, 75-79, 143-149)
# ----------------------------------------------------------------------------------------


def num_hits(num_results: int, oversample: int = 4, chunk_max_size: int = 2048) -> int:
    """`round(oversample * chunk_max_size / 2048) * max(num_results, 10)` (`_search.py:66-67`)."""
    corrected_oversample = oversample * chunk_max_size / 2048
    return round(corrected_oversample) * max(num_results, 10)


def topk_desc(scores: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Exact top-k by (score descending, index ascending).  NaNs rank last.

    `ORDER BY dist LIMIT k` (`_search.py:77-79`) with the tie order SQL leaves unspecified
    fixed to the lowest index; NaN ranks after -inf."""
    scores = np.asarray(scores)
    k = min(k, len(scores))
    nan = np.isnan(scores)
    key = np.where(nan, -np.inf, scores)
    order = np.lexsort((np.arange(len(scores)), -key, nan))[:k]  # NaN strictly after -inf
    return scores[order], order.astype(np.int64)


def search_rows(E, q, k, metric="cosine", dtype=np.float64):
    """a6+a7: top-k rows by similarity."""
    return topk_desc(similarity(E, q, metric, dtype), k)


def group_chunk_max(row_scores: np.ndarray, row_chunks: np.ndarray, num_results: int):
    """a8: `SELECT chunk_id, max(sim) GROUP BY chunk_id ORDER BY max DESC LIMIT n`
    (`_search.py:143-149`) over the a7 rows.  Ties -> lowest chunk ordinal."""
    best: dict[int, float] = {}
    for s, c in zip(row_scores.tolist(), row_chunks.tolist()):
        if c not in best or s > best[c]:
            best[c] = s
    items = sorted(best.items(), key=lambda kv: (-kv[1], kv[0]))[:num_results]
    return (
        np.asarray([s for _, s in items], dtype=row_scores.dtype),
        np.asarray([c for c, _ in items], dtype=np.int64),
    )


def search_chunks(E, row_to_chunk, q, n_hits, num_results, metric="cosine", dtype=np.float64):
    """a6+a7+a8: the reference's two-stage semantics (rows -> per-chunk max -> chunks).
    Fewer than `num_results` chunks come back when the `n_hits` best rows span fewer chunks."""
    s, rows = search_rows(E, q, n_hits, metric, dtype)
    return group_chunk_max(s, np.asarray(row_to_chunk)[rows], num_results)


# ----------------------------------------------------------------------------------------
# 8f-1: metadata filter (filter-first branch) and deleted chunks
# ----------------------------------------------------------------------------------------


def search_chunks_filtered(E, row_to_chunk, q, n_hits, num_results, chunk_ok, metric="cosine", dtype=np.float64):
    """The reference's filtered vector search (`_search.py:96-149`) for an exact engine.

    `chunk_ok[c]` = chunk c matches the metadata filter (`json_contains`, `:88-94`) and still exists
    (deleted documents' chunks are gone from the table, `_delete.py:148-176`).  Filter-first branch
    (`:105-119`): `WHERE chunk_id IN (matching) ORDER BY dist LIMIT n_hits`, then the per-chunk max
    (`:143-149`).  The other branch (`:120-141`, more than 100 000 matching rows) first cuts the table
    to its 1 000 000 nearest rows -- an artefact of serving it from the ANN index; on tables of up to
    1 000 000 rows (every BASELINE config's single-GPU shard) both branches return the same rows, and
    the restatement ranks exactly over all matching rows.
    Returns (scores, chunk ordinals) like `group_chunk_max`."""
    r2c = np.asarray(row_to_chunk)
    ok_rows = np.asarray(chunk_ok, dtype=bool)[r2c]
    sims = np.where(ok_rows, similarity(E, q, metric, dtype), -np.inf)
    s, rows = topk_desc(sims, n_hits)
    keep = rows >= 0
    keep[keep] &= ok_rows[rows[keep]]  # non-matching rows only surface when fewer than n_hits rows match
    return group_chunk_max(s[keep], r2c[rows[keep]], num_results)


def search_rows_ranked(E, row_to_chunk, q, k, chunk_ok, rank_limit, live_chunk=None, metric="cosine", dtype=np.float64):
    """The order-first-then-filter branch of the reference (`_search.py:120-141`, taken when the filter matches more than
    100 000 rows): `ORDER BY dist LIMIT rank_limit` over the unfiltered table (`:121-126`, rank_limit = 1 000 000 there), the
    metadata filter on those rows (`:127-139`), `ORDER BY dist LIMIT k` (`:138-139`).  `live_chunk[c]` = chunk c is still in
    the table (`_delete.py:148-176`); ties are resolved to the lowest row (SQL leaves them unspecified).
    Unfilled slots are (-inf, -1)."""
    r2c = np.asarray(row_to_chunk)
    n = len(r2c)
    live_rows = np.ones(n, dtype=bool) if live_chunk is None else np.asarray(live_chunk, dtype=bool)[r2c]
    ok_rows = np.asarray(chunk_ok, dtype=bool)[r2c] & live_rows
    sims = similarity(E, q, metric, dtype)
    _, nearest = topk_desc(np.where(live_rows, sims, -np.inf), min(int(rank_limit), n))
    eligible = np.zeros(n, dtype=bool)
    eligible[nearest[nearest >= 0]] = True
    eligible &= ok_rows
    s, rows = topk_desc(np.where(eligible, sims, -np.inf), k)
    dead = (rows >= 0) & ~eligible[np.clip(rows, 0, max(n - 1, 0))] if n else rows >= 0
    return np.where(dead, -np.inf, s), np.where(dead, -1, rows)


def search_chunks_ranked(E, row_to_chunk, q, n_hits, num_results, chunk_ok, rank_limit, live_chunk=None, metric="cosine",
                         dtype=np.float64):
    """`search_rows_ranked` followed by the per-chunk max of `_search.py:143-149`."""
    r2c = np.asarray(row_to_chunk)
    s, rows = search_rows_ranked(E, r2c, q, n_hits, chunk_ok, rank_limit, live_chunk, metric, dtype)
    keep = rows >= 0
    return group_chunk_max(s[keep], r2c[rows[keep]], num_results)


def search_rows_filtered(E, row_to_chunk, q, k, chunk_ok, metric="cosine", dtype=np.float64):
    """Top-k rows among the rows of matching chunks; unfilled slots are (-inf, -1)."""
    r2c = np.asarray(row_to_chunk)
    ok_rows = np.asarray(chunk_ok, dtype=bool)[r2c]
    sims = np.where(ok_rows, similarity(E, q, metric, dtype), -np.inf)
    s, rows = topk_desc(sims, k)
    dead = (rows >= 0) & ~ok_rows[np.clip(rows, 0, max(len(ok_rows) - 1, 0))] if len(ok_rows) else rows >= 0
    return np.where(dead, -np.inf, s), np.where(dead, -1, rows)


def maxsim_topk_filtered(D, chunk_offsets, Q, k, chunk_ok, dtype=np.float64):
    """MaxSim top-k among matching chunks; unfilled slots are (-inf, -1)."""
    sc = maxsim_scores(D, chunk_offsets, Q, dtype)
    ok = np.asarray(chunk_ok, dtype=bool)
    s, c = topk_desc(np.where(ok, sc, -np.inf), k)
    dead = (c >= 0) & (~ok[np.clip(c, 0, max(len(ok) - 1, 0))] | np.isneginf(s))
    return np.where(dead, -np.inf, s), np.where(dead, -1, c)


# ----------------------------------------------------------------------------------------
# 8f-3: the query adapter's producer (`_query_adapter.py:20-38,153-205`) -- pinned by
# tests/golden/query_adapter.npz (oracle/make_golden_adapter.py executes the reference's own lines)
# ----------------------------------------------------------------------------------------


def best_row(E_chunk: np.ndarray, q: np.ndarray) -> int:
    """`np.argmax(chunk.embedding_matrix @ q)` (`_query_adapter.py:174,180`): first maximum on ties."""
    return int(np.argmax(np.asarray(E_chunk) @ np.asarray(q)))


def optimize_query_target(q: np.ndarray, P: np.ndarray, N: np.ndarray, alpha: float = 0.05) -> np.ndarray:  # noqa: N803
    """`_optimize_query_target` (`_query_adapter.py:20-38`): t* = q + Dᵀ μ*, μ* = argmin ½‖q + Dᵀμ‖², μ ≥ 0,
    D = all P_i − (1 + α) N_j; solved in fp64, cast back to q's dtype."""
    from scipy.optimize import lsq_linear

    dt = q.dtype
    q64, P64, N64 = q.astype(np.float64), P.astype(np.float64), N.astype(np.float64)
    D = np.reshape(P64[:, np.newaxis, :] - (1.0 + alpha) * N64[np.newaxis, :, :], (-1, P64.shape[1]))
    mu = lsq_linear(D.T, -q64, bounds=(0.0, np.inf), tol=np.finfo(np.float64).eps).x
    return (q64 + D.T @ mu).astype(dt)


def query_adapter_from_targets(Q: np.ndarray, T: np.ndarray, metric: str = "cosine") -> np.ndarray:  # noqa: N803
    """`_query_adapter.py:182-205`: normalise the rows of Q (and of T for cosine), M = TᵀQ / n completed on Q's null
    space, then the orthogonal Procrustes solution U Vᵀ (cosine) or M scaled to Frobenius norm √d (dot)."""
    Q = np.array(Q, dtype=np.float64)
    T = np.array(T, dtype=np.float64)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if metric == "cosine":
        T /= np.linalg.norm(T, axis=1, keepdims=True)
    n, d = Q.shape
    M = (1 / n) * T.T @ Q
    if n < d or np.linalg.matrix_rank(Q) < d:
        M += np.eye(d) - Q.T @ np.linalg.pinv(Q @ Q.T) @ Q
    if metric == "dot":
        return M / np.linalg.norm(M, ord="fro") * np.sqrt(d)
    if metric == "cosine":
        U, _, VT = np.linalg.svd(M, full_matrices=False)
        return U @ VT
    raise ValueError(f"Unsupported metric: {metric}")


def update_query_adapter(evals, E, chunk_offsets, chunk_ids, *, optimize_top_k=40, optimize_gap=0.05, metric="cosine",
                         oversample=4, chunk_max_size=2048, dtype=np.float64):
    """The whole loop of `update_query_adapter` (`_query_adapter.py:153-205`) over an in-memory table.

    evals: sequence of (query vector, relevant chunk ids).  Per eval: vector search WITHOUT the adapter for
    `optimize_top_k` chunks (`:166-168`), skip unless both relevant and irrelevant chunks were retrieved (`:171-173`),
    P / N = best row of every relevant / irrelevant retrieved chunk (`:174-181`), target t (`:183`).  Returns
    (A*, Q, T) with Q, T the stacked un-normalised rows."""
    off = np.asarray(chunk_offsets, dtype=np.int64)
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    E = np.asarray(E)
    Qs, Ts = [], []
    for q, relevant in evals:
        q = np.asarray(q)
        n_hits = num_hits(optimize_top_k, oversample, chunk_max_size)
        _, chunks = search_chunks(E, r2c, q.astype(dtype), n_hits, optimize_top_k, metric, dtype)
        rel = np.asarray([chunk_ids[c] in relevant for c in chunks], dtype=bool)
        if not rel.any() or rel.all():
            continue
        rows = [int(off[c]) + best_row(E[off[c] : off[c + 1]], q) for c in chunks]
        P = E[[r for r, ok in zip(rows, rel) if ok]]
        N = E[[r for r, ok in zip(rows, rel) if not ok]]
        Ts.append(optimize_query_target(q, P, N, optimize_gap))
        Qs.append(q)
    if not Qs:
        raise ValueError("no eval retrieved both relevant and irrelevant chunks")
    Q, T = np.vstack(Qs).astype(np.float64), np.vstack(Ts).astype(np.float64)
    return query_adapter_from_targets(Q, T, metric), Q, T


# ----------------------------------------------------------------------------------------
# 8f-4: semantic-chunking similarities (`_split_chunks.py:54-86`) -- pinned by
# tests/golden/split_chunks.npz (oracle/make_golden_chunks.py runs the reference's own function)
# ----------------------------------------------------------------------------------------


def nonoutlying_mask(chunklet_size: np.ndarray) -> np.ndarray:
    """`_split_chunks.py:57-58`: chunklets whose size lies within the 15 %..85 % quantiles."""
    q15, q85 = np.quantile(chunklet_size, [0.15, 0.85])
    return (q15 <= chunklet_size) & (chunklet_size <= q85)


def partition_similarity(chunklet_embeddings: np.ndarray, chunklet_size: np.ndarray) -> np.ndarray:
    """`_split_chunks.py:54-72` in float32 like the reference: unit rows, discourse vector removed unless that
    would zero a row, similarity of consecutive rows mapped to ((s + 1) / 2) and floored at sqrt(eps)."""
    X = chunklet_embeddings.astype(np.float32)
    X = X / np.linalg.norm(X, axis=1, keepdims=True)
    sel = nonoutlying_mask(np.asarray(chunklet_size))
    if np.any(sel):
        discourse = np.mean(X[sel, :], axis=0)
        discourse = discourse / np.linalg.norm(discourse)
        X_mod = X - np.outer(X @ discourse, discourse)
        if not np.any(np.linalg.norm(X_mod, axis=1) <= np.finfo(X.dtype).eps):
            X = X_mod / np.linalg.norm(X_mod, axis=1, keepdims=True)
    sim = np.sum(X[:-1] * X[1:], axis=1)
    return np.maximum((sim + 1) / 2, np.sqrt(np.finfo(X.dtype).eps))


def heading_adjusted(sim: np.ndarray, chunklets: list[str]) -> np.ndarray:
    """`_split_chunks.py:73-86`: splitting before a Markdown heading is encouraged (/4), right after one forbidden (1.0)."""
    import re

    sim = np.array(sim, copy=True)
    prev_is_heading = True
    for i, c in enumerate(chunklets[:-1]):
        is_heading = bool(re.match(r"^#+\s", c.replace("\n", "").strip()))
        if is_heading:
            if not prev_is_heading:
                sim[i - 1] = sim[i - 1] / 4
            sim[i] = 1.0
        prev_is_heading = is_heading
    return sim


# ----------------------------------------------------------------------------------------
# a9: MaxSim (multi-query generalisation of _search.py:143-149 / _query_adapter.py:174)
# ----------------------------------------------------------------------------------------


def maxsim_scores(D: np.ndarray, chunk_offsets: np.ndarray, Q: np.ndarray, dtype=np.float64) -> np.ndarray:
    """score[c] = sum_i max_{j in chunk c} Q[i] . D[j]; empty chunks score -inf."""
    D = np.asarray(D, dtype=dtype)
    Q = np.atleast_2d(np.asarray(Q, dtype=dtype))
    off = np.asarray(chunk_offsets, dtype=np.int64)
    n_chunks = len(off) - 1
    out = np.full(n_chunks, -np.inf, dtype=dtype)
    nonempty = np.nonzero(off[1:] > off[:-1])[0]
    if len(nonempty) == 0 or len(D) == 0:
        return out
    S = D @ Q.T  # (N, nq)
    seg_max = np.maximum.reduceat(S, off[nonempty], axis=0)  # valid because nonempty starts ascend
    out[nonempty] = seg_max.sum(axis=1)
    return out


def maxsim_topk(D, chunk_offsets, Q, k, dtype=np.float64):
    return topk_desc(maxsim_scores(D, chunk_offsets, Q, dtype), k)


def maxsim_scores_batch(D: np.ndarray, chunk_offsets: np.ndarray, Qb: np.ndarray, dtype=np.float64,
                        slab_rows: int = 65536) -> np.ndarray:
    """`maxsim_scores` for a batch of queries Qb (n_queries, nq, dim) -> (n_queries, n_chunks), same arithmetic per
    (row, query vector) product: the corpus goes through in chunk-aligned row slabs with ONE matrix product per slab
    for all queries' vectors (what makes the CPU baseline use its BLAS threads well), then the per-chunk maximum
    (`np.maximum.reduceat`, the NumPy analogue of `max(sim) GROUP BY chunk_id`, `_search.py:143-149`) and the sum over
    each query's vectors.  No empty chunks (callers with empty chunks use `maxsim_scores`)."""
    off = np.asarray(chunk_offsets, dtype=np.int64)
    n_chunks = len(off) - 1
    Qb = np.asarray(Qb, dtype=dtype)
    n_queries, nq, dim = Qb.shape
    assert np.all(off[1:] > off[:-1]), "maxsim_scores_batch: empty chunk"
    Qall = np.ascontiguousarray(Qb.reshape(n_queries * nq, dim).T)  # (dim, n_queries * nq)
    out = np.empty((n_queries, n_chunks), dtype=dtype)
    c0 = 0
    while c0 < n_chunks:
        c1 = int(np.searchsorted(off, off[c0] + slab_rows, side="right")) - 1
        c1 = min(max(c1, c0 + 1), n_chunks)
        r0, r1 = int(off[c0]), int(off[c1])
        S = np.asarray(D[r0:r1], dtype=dtype) @ Qall  # (rows, n_queries * nq)
        seg = np.maximum.reduceat(S, off[c0:c1] - r0, axis=0)  # (chunks, n_queries * nq)
        out[:, c0:c1] = seg.reshape(c1 - c0, n_queries, nq).sum(axis=2).T
        c0 = c1
    return out


def maxsim_candidates(D, chunk_offsets, Q, cand, dtype=np.float64) -> np.ndarray:
    """MaxSim restricted to candidate chunk ordinals (rerank shape, SURVEY cfg 3)."""
    D = np.asarray(D, dtype=dtype)
    Q = np.atleast_2d(np.asarray(Q, dtype=dtype))
    off = np.asarray(chunk_offsets, dtype=np.int64)
    out = np.empty(len(cand), dtype=dtype)
    for i, c in enumerate(cand):
        b, e = off[c], off[c + 1]
        out[i] = (D[b:e] @ Q.T).max(axis=0).sum() if e > b else -np.inf
    return out


# ----------------------------------------------------------------------------------------
# section 8e: shard merge
# ----------------------------------------------------------------------------------------


def merge_topk(scores_list: list[np.ndarray], ids_list: list[np.ndarray], k: int):
    """Concatenate per-shard top-k lists (global ids) and take the global top-k by
    (score desc, id asc): identical to the single-shard result."""
    s = np.concatenate(scores_list)
    i = np.concatenate(ids_list).astype(np.int64)
    nan = np.isnan(s)
    key = np.where(nan, -np.inf, s)
    order = np.lexsort((i, -key, nan))[:k]
    return s[order], i[order]


def shard_bounds_by_chunk(chunk_offsets: np.ndarray, world: int) -> list[tuple[int, int]]:
    """Contiguous chunk ranges per rank, balanced by row count, chunks never split."""
    off = np.asarray(chunk_offsets, dtype=np.int64)
    n_rows, n_chunks = int(off[-1]), len(off) - 1
    cuts = [0]
    for r in range(1, world):
        target = (n_rows * r) // world
        c = int(np.searchsorted(off, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), n_chunks))
    cuts.append(n_chunks)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]
