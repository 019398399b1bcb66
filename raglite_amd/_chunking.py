"""Host mirror of `src/raglite/_split_chunks.py` with the similarity arithmetic on the GPU.

    split_chunks(chunklets, chunklet_embeddings, max_size=2048) -> (chunks, chunk_embeddings)   (`_split_chunks.py:13-122`)

This is the step between a1-a3 (pooled chunklet embeddings) and a4 (the contiguous row spans that make a chunk a
multi-vector object, `np.split(chunklet_embeddings, partition_indices)`).  What runs on the device
(`rl_partition_similarity`, batched over documents when called through `partition_similarities`): row normalisation,
discourse-vector removal and the similarity of consecutive chunklets (`:54-72`).  What stays on the host: string
lengths and their quantiles, the Markdown-heading adjustments (`:73-86`) and the binary integer programme (`:87-113`,
scipy's HiGHS, as in the reference).
"""

from __future__ import annotations

import ctypes as C
import re
from typing import Any, Sequence

import numpy as np

from raglite_amd import _ops
from raglite_amd._abi import check, lib

_HEADING = re.compile(r"^#+\s")


def _nonoutlying(sizes: np.ndarray) -> np.ndarray:
    q15, q85 = np.quantile(sizes, [0.15, 0.85])  # `_split_chunks.py:57-58`
    return ((q15 <= sizes) & (sizes <= q85)).astype(np.uint8)


def partition_similarities(embeddings: Any, doc_offsets: np.ndarray, chunklet_sizes: np.ndarray) -> Any:
    """Similarities of consecutive chunklets for MANY documents in one launch.

    embeddings: (N, dim) NumPy array or CUDA tensor, the documents' chunklet embeddings concatenated;
    doc_offsets: int64[n_docs + 1]; chunklet_sizes: int64[N] string lengths.  Returns float32[N] on the same side as
    `embeddings` (entry i = chunklets i and i+1 of one document; 0 at every document's last chunklet)."""
    off = np.ascontiguousarray(doc_offsets, dtype=np.int64)
    sizes = np.asarray(chunklet_sizes)
    sel = np.concatenate([_nonoutlying(sizes[off[d] : off[d + 1]]) if off[d + 1] > off[d] else np.zeros(0, np.uint8)
                          for d in range(len(off) - 1)]) if len(off) > 1 else np.zeros(0, np.uint8)
    a = _ops._Args()  # noqa: SLF001
    p_x = a.inp(embeddings, np.float32)
    x = a.keep[0]
    n, dim = int(x.shape[0]), int(x.shape[1])
    if n != int(off[-1]) or len(sel) != n:
        raise ValueError("doc_offsets / chunklet_sizes do not match the embedding rows")
    if a.mem == _ops.MEM_DEVICE:
        torch = _ops._torch()  # noqa: SLF001
        t_off = torch.from_numpy(off).to(a.device)
        t_sel = torch.from_numpy(sel).to(a.device)
        a.keep += [t_off, t_sel]
        p_off, p_sel = t_off.data_ptr(), t_sel.data_ptr()
    else:
        a.keep += [off, sel]
        p_off, p_sel = off.ctypes.data, sel.ctypes.data
    out, p_out = a.out((n,), np.float32)
    a.ensure_device()
    check(lib().rl_partition_similarity(p_x, n, dim, p_off, len(off) - 1, p_sel, p_out, a.mem, a.stream))
    return out


def _apply_headings(sim: np.ndarray, chunklets: Sequence[str]) -> np.ndarray:
    """`_split_chunks.py:73-86`."""
    prev_is_heading = True
    for i, chunklet in enumerate(chunklets[:-1]):
        is_heading = bool(_HEADING.match(chunklet.replace("\n", "").strip()))
        if is_heading:
            if not prev_is_heading:
                sim[i - 1] = sim[i - 1] / 4  # encourage a split before the heading
            sim[i] = 1.0  # never split right after it
        prev_is_heading = is_heading
    return sim


def _solve_partition(cost: np.ndarray, sizes: np.ndarray, max_size: int) -> list[int]:
    """`_split_chunks.py:87-113`: minimise cost . x over binary x (x[i] = split after chunklet i) such that every
    window of chunklets that overflows `max_size` contains a split."""
    from scipy.optimize import linprog
    from scipy.sparse import coo_matrix

    csum = np.cumsum(sizes)
    starts = np.concatenate(([0], csum[:-1]))
    n = len(sizes)
    # first chunklet index (exclusive end) whose inclusion overflows a chunk starting at i
    ends = np.searchsorted(csum, starts[: n - 1] + max_size, side="right")
    rows_needed = int(np.argmax(ends == n)) if np.any(ends == n) else n - 1  # the reference stops at the first fit
    rows, cols = [], []
    for i in range(rows_needed):
        span = np.arange(i, ends[i])
        rows.append(np.full(len(span), i))
        cols.append(span)
    if not rows:
        return []
    A = coo_matrix((np.ones(sum(len(r) for r in rows), np.float32), (np.concatenate(rows), np.concatenate(cols))),  # noqa: N806
                   shape=(rows_needed, n - 1), dtype=np.float32)
    res = linprog(cost, A_ub=-A, b_ub=-np.ones(A.shape[0], np.float32), bounds=(0, 1), integrality=[1] * A.shape[1])
    if not res.success:
        raise ValueError("Optimization of chunk partitions failed.")
    return (np.where(res.x)[0] + 1).tolist()


def partition_cost(chunklets: Sequence[str], chunklet_embeddings: Any) -> np.ndarray:
    """The MILP's cost vector for one document: device similarities + host heading adjustments (float32[n - 1])."""
    sizes = np.asarray([len(c) for c in chunklets])
    sim = partition_similarities(chunklet_embeddings, np.asarray([0, len(chunklets)], np.int64), sizes)
    sim = sim.cpu().numpy() if hasattr(sim, "cpu") else np.asarray(sim)
    return _apply_headings(sim[:-1].astype(np.float32), chunklets)


def split_chunks(chunklets: list[str], chunklet_embeddings: Any, max_size: int = 2048) -> tuple[list[str], list[Any]]:
    """Split chunklets into optimal semantic chunks (same contract and error messages as the reference)."""
    sizes = np.asarray([len(c) for c in chunklets])
    if not np.all(sizes <= max_size):
        raise ValueError("Chunklet larger than chunk max_size detected.")
    emb_host = chunklet_embeddings.float().cpu().numpy() if hasattr(chunklet_embeddings, "cpu") else np.asarray(chunklet_embeddings)
    if not np.all(np.linalg.norm(emb_host.astype(np.float32), axis=1) > 0.0):
        raise ValueError("Chunklet embeddings with zero norm detected.")
    if len(chunklets) <= 1 or int(sizes.sum()) <= max_size:
        return ["".join(chunklets)] if chunklets else chunklets, [chunklet_embeddings]
    cost = partition_cost(chunklets, chunklet_embeddings)
    cuts = _solve_partition(cost, sizes, max_size)
    bounds = [0, *cuts, len(chunklets)]
    chunks = ["".join(chunklets[i:j]) for i, j in zip(bounds[:-1], bounds[1:])]
    if hasattr(chunklet_embeddings, "cpu"):
        parts = [chunklet_embeddings[i:j] for i, j in zip(bounds[:-1], bounds[1:])]
    else:
        parts = np.split(np.asarray(chunklet_embeddings), cuts)  # `_split_chunks.py:121`
    return chunks, list(parts)
