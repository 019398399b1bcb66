"""Host mirror of `src/raglite/_query_adapter.py` with the searches batched on the GPU.

    update_query_adapter(evals, *, max_evals=4096, optimize_top_k=40, optimize_gap=0.05, config=None, index=None)
        -> (d, d) float64 adapter, also installed as `index.query_adapter`              (`_query_adapter.py:41-219`)

The reference loops over the evals: embed the question, run one vector search WITHOUT the adapter, pick the best row
of every retrieved chunk as positive / negative example, solve a small NNLS for the target vector, and finally fit
the adapter in closed form.  Here every eval's search goes through ONE `rl_search_chunks` call (the B = hundreds-to-
thousands batched shape of BASELINE cfg 5, i.e. the fp32 MFMA GEMM path), the best rows through ONE
`rl_chunk_best_rows`, the example rows come back through ONE `rl_gather_rows`; the NNLS (`scipy.optimize.lsq_linear`)
and the Procrustes / pseudo-inverse algebra stay on the host in fp64 exactly as written in the reference.  Evals come
from the store in the reference (`select(Eval)`, `:150`); here they are passed in.
"""

from __future__ import annotations

from dataclasses import replace
from typing import Any, Sequence

import numpy as np

from raglite_amd import _ops
from raglite_amd._config import DEFAULT_CHUNK_MAX_SIZE, HotPathConfig
from raglite_amd._embed import embed_strings, embedding_type
from raglite_amd._search import GpuIndex, _index_for


def _optimize_query_target(q: np.ndarray, P: np.ndarray, N: np.ndarray, *, alpha: float = 0.05) -> np.ndarray:  # noqa: N803
    """`_query_adapter.py:20-38`: t* = q + Dᵀ μ*, μ* = argmin ½‖q + Dᵀ μ‖² s.t. μ ≥ 0, D = P_i − (1 + α) N_j."""
    from scipy.optimize import lsq_linear

    q_dtype = q.dtype
    q, P, N = q.astype(np.float64), P.astype(np.float64), N.astype(np.float64)  # noqa: N806
    D = np.reshape(P[:, np.newaxis, :] - (1.0 + alpha) * N[np.newaxis, :, :], (-1, P.shape[1]))  # noqa: N806
    mu = lsq_linear(D.T, -q, bounds=(0.0, np.inf), tol=np.finfo(np.float64).eps).x
    return (q + D.T @ mu).astype(q_dtype)


def _adapter_from_targets(Q: np.ndarray, T: np.ndarray, metric: str) -> np.ndarray:  # noqa: N803
    """`_query_adapter.py:182-208`."""
    Q = Q / np.linalg.norm(Q, axis=1, keepdims=True)  # noqa: N806
    if metric == "cosine":
        T = T / np.linalg.norm(T, axis=1, keepdims=True)  # noqa: N806
    n, d = Q.shape
    M = (1 / n) * T.T @ Q  # noqa: N806
    if n < d or np.linalg.matrix_rank(Q) < d:
        M += np.eye(d) - Q.T @ np.linalg.pinv(Q @ Q.T) @ Q  # noqa: N806
    if metric == "dot":
        return M / np.linalg.norm(M, ord="fro") * np.sqrt(d)
    if metric == "cosine":
        U, _, VT = np.linalg.svd(M, full_matrices=False)  # noqa: N806
        return U @ VT
    raise ValueError(f"Unsupported metric: {metric}")


def _eval_fields(ev: Any) -> tuple[Any, Sequence[str]]:
    if isinstance(ev, (tuple, list)):
        return ev[0], ev[1]
    return ev.question, ev.chunk_ids  # the reference's `Eval` row (`_database.py`, fields question / chunk_ids)


def update_query_adapter(evals: Sequence[Any], *, max_evals: int = 4096, optimize_top_k: int = 40,
                         optimize_gap: float = 0.05, oversample: int = 4, config: Any | None = None,
                         index: GpuIndex | None = None) -> np.ndarray:
    """Compute the optimal query adapter from evals and install it on the index.

    evals: `Eval`-like objects (`.question`, `.chunk_ids`) or `(question, chunk_ids)` pairs; a question may be a
    string (embedded with `embed_strings`) or an already embedded vector."""
    config = config or HotPathConfig()
    gi = index or _index_for(config)
    if gi.index.n_rows == 0:
        raise ValueError("First run `insert_documents()` to insert documents.")  # `_query_adapter.py:146-148`
    evals = list(evals)[:max_evals]
    if len(evals) == 0:
        raise ValueError("First run `insert_evals()` to generate evals.")  # `:150-152`
    metric = config.vector_search_distance_metric
    if metric not in ("cosine", "dot"):
        raise ValueError(f"Unsupported metric: {metric}")  # `:206-208` (checked up front: nothing is computed for l2)
    no_adapter = replace(config, vector_search_query_adapter=False) if hasattr(config, "__dataclass_fields__") else config
    # ---- embed the questions (`:160`) -----------------------------------------------------------------------
    fields = [_eval_fields(ev) for ev in evals]
    texts = [q for q, _ in fields if isinstance(q, str)]
    # A late-chunking embedder treats a LIST of strings as the sentences of one document (they are joined and pooled
    # with each other's context); the reference embeds each question alone (`embed_strings([eval_.question])[0]`,
    # `_query_adapter.py:160`) and so does `vector_search` at query time.  Only the standard embedder is batched.
    if texts and embedding_type(config=no_adapter) == "late_chunking":
        embedded = iter([embed_strings([t], config=no_adapter)[0] for t in texts])
    else:
        embedded = iter(embed_strings(texts, config=no_adapter)) if texts else iter(())
    qs = [next(embedded) if isinstance(q, str) else np.ravel(np.asarray(q)) for q, _ in fields]
    Q_all = np.vstack([np.asarray(q, dtype=np.float32) for q in qs])  # noqa: N806
    # ---- ONE batched search without the adapter (`:162-165`, num_hits as in `_search.py:66-67`) --------------
    chunk_max_size = getattr(config, "chunk_max_size", DEFAULT_CHUNK_MAX_SIZE)
    num_hits = round(oversample * chunk_max_size / 2048) * max(optimize_top_k, 10)
    k = min(optimize_top_k, _ops.K_MAX)
    _, chunks, counts = gi.index.search_chunks(Q_all, min(num_hits, _ops.K_MAX), k)
    chunks = np.asarray(chunks).reshape(len(qs), k)
    counts = np.asarray(counts).reshape(len(qs))
    # ---- best row of every retrieved chunk (`:174,180`), ONE launch ------------------------------------------
    best = np.asarray(gi.index.chunk_best_rows(Q_all, chunks.astype(np.int32))).reshape(len(qs), k)
    # ---- which evals qualify (`:168-173`) --------------------------------------------------------------------------
    keep, rel_masks = [], []
    for i, (_, relevant) in enumerate(fields):
        n = int(counts[i])
        relevant = set(relevant)
        rel = np.fromiter((gi.chunk_ids[c] in relevant for c in chunks[i, :n]), dtype=bool, count=n)
        if rel.any() and not rel.all():
            keep.append(i)
            rel_masks.append(rel)
    if not keep:
        raise ValueError("No eval retrieved both relevant and irrelevant chunks; cannot fit a query adapter.")
    # ---- fetch the example rows in one go, solve the per-eval NNLS on the host (`:183`) ------------------------
    wanted = np.concatenate([best[i, : len(rel)] for i, rel in zip(keep, rel_masks)]).astype(np.int32)
    rows = np.asarray(gi.index.gather_rows(wanted))
    Q_rows, T_rows, base = [], [], 0  # noqa: N806
    for i, rel in zip(keep, rel_masks):
        ex = rows[base : base + len(rel)].astype(qs[i].dtype)  # `Chunk.embedding_matrix` has the stored dtype
        base += len(rel)
        q = qs[i]
        T_rows.append(_optimize_query_target(q, ex[rel], ex[~rel], alpha=optimize_gap))
        Q_rows.append(q)
    A_star = _adapter_from_targets(np.vstack(Q_rows).astype(np.float64), np.vstack(T_rows).astype(np.float64), metric)  # noqa: N806
    gi.query_adapter = np.asarray(A_star, dtype=np.float32)  # `IndexMetadata["query_adapter"]` (`:210-214`)
    return A_star
