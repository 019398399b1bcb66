"""Read the device index straight from a RAGLite database.

The reference keeps one row per chunklet vector in `chunk_embedding` (`src/raglite/_database.py:403-430`: `id`
auto-increment, `chunk_id` -> `chunk.id`, `embedding`) and the chunks' text and metadata in `chunk`
(`:207-224`: `id`, `document_id`, `index`, `headings`, `body`, `metadata` JSON).  `Chunk.embedding_matrix`
(`:279-283`) stacks a chunk's rows in relationship order; the device image wants them ordered by
(`chunk_id`, `id`), all rows of a chunk contiguous.  This module is plain SQLAlchemy Core -- textual SELECTs over
those columns -- so it works against any store the reference can create (SQLite, DuckDB, PostgreSQL) without
importing the reference's ORM classes:

  embedding column   SQLite / generic: `np.save` bytes (`_typing.py:57-78`); DuckDB: FLOAT[d] list
                     (`:178-208`); PostgreSQL: halfvec text "[..]" (`:145-175`)
  index_metadata     `query_adapter` lives in the pickled `metadata` of row "default" (`_database.py:434-462`)
"""

from __future__ import annotations

import io
import json
import pickle
from dataclasses import dataclass, field
from typing import Any, Iterable, Sequence

import numpy as np


def _connection(bind: Any):
    """(connection, owned) for an Engine, Connection, ORM Session or database URL."""
    import sqlalchemy as sa
    from sqlalchemy.engine import Connection, Engine

    if isinstance(bind, str):
        return sa.create_engine(bind).connect(), True
    if isinstance(bind, Engine):
        return bind.connect(), True
    if isinstance(bind, Connection):
        return bind, False
    if hasattr(bind, "connection") and callable(bind.connection):  # ORM Session (sqlmodel / sqlalchemy.orm)
        return bind.connection(), False
    raise TypeError(f"cannot read a RAGLite store from {type(bind).__name__}: pass an Engine, Connection, Session or URL")


def decode_embedding(value: Any) -> np.ndarray:
    """One `chunk_embedding.embedding` value, whatever the dialect stored, as a float32 vector."""
    if value is None:
        raise ValueError("chunk_embedding.embedding is NULL")
    if isinstance(value, (bytes, bytearray, memoryview)):
        return np.asarray(np.load(io.BytesIO(bytes(value)), allow_pickle=False), dtype=np.float32).ravel()
    if isinstance(value, str):  # halfvec text: the reference parses it as float16 (`_typing.py:165-175`), which is what was stored
        return np.asarray(value.strip("[]").split(","), dtype=np.float64).astype(np.float16).astype(np.float32)
    return np.asarray(value, dtype=np.float32).ravel()


def chunk_text(headings: str, body: str, metadata: dict) -> str:
    """`str(chunk)` of the reference (`_database.py:300-324`): YAML-ish front matter from filename / url / uri, the
    contextual headings, the body -- what `rerank_chunks` hands to a reranker (`_search.py:394-396`).  Metadata values are
    formatted as stored: the reference keeps every value as a list (`_database.py:51-55`), so a chunk of `a.md` reads
    `filename: ['a.md']`, and `url: [None]` is a (truthy) line of its own."""
    lines = "\n".join(f"{key}: {metadata.get(key)}" for key in ("filename", "url", "uri") if metadata.get(key))
    front = f"---\n{lines}\n---" if lines else ""
    return f"{front}\n\n{(headings or '').strip()}\n\n{(body or '').strip()}".strip()


def _as_dict(value: Any) -> dict:
    if value is None:
        return {}
    if isinstance(value, (bytes, bytearray)):
        value = bytes(value).decode()
    if isinstance(value, str):
        return json.loads(value) if value else {}
    return dict(value)


@dataclass
class StoreImage:
    chunk_ids: list[str] = field(default_factory=list)
    sizes: list[int] = field(default_factory=list)           # rows per chunk
    rows: list[np.ndarray] = field(default_factory=list)     # one float32 vector per chunk_embedding row, in order
    docs: list[str] = field(default_factory=list)
    metadata: list[dict] = field(default_factory=list)
    query_adapter: np.ndarray | None = None

    def matrix(self) -> np.ndarray:
        return np.vstack(self.rows).astype(np.float32, copy=False) if self.rows else np.zeros((0, 1), np.float32)


def _in_batches(items: Sequence[str], n: int = 500) -> Iterable[Sequence[str]]:
    for i in range(0, len(items), n):
        yield items[i : i + n]


def read_chunks(conn, only_chunk_ids: Sequence[str] | None = None) -> StoreImage:
    """Embedding rows ordered by (chunk_id, id) with their chunk's text and metadata; chunks without embeddings are not
    part of the vector index.  `only_chunk_ids`: restrict to these chunks (incremental sync)."""
    import sqlalchemy as sa

    img = StoreImage()

    def fetch(where: str, params: dict):
        emb = conn.execute(sa.text(f"SELECT chunk_id, embedding FROM chunk_embedding {where} ORDER BY chunk_id, id"), params)
        last = None
        for chunk_id, value in emb:
            if chunk_id != last:
                img.chunk_ids.append(chunk_id)
                img.sizes.append(0)
                last = chunk_id
            img.sizes[-1] += 1
            img.rows.append(decode_embedding(value))
        meta = {}
        rows = conn.execute(sa.text(f'SELECT id, headings, body, metadata FROM chunk {where.replace("chunk_id", "id")}'), params)
        for cid, headings, body, md in rows:
            meta[cid] = (headings, body, _as_dict(md))
        return meta

    if only_chunk_ids is None:
        meta = fetch("", {})
    else:
        meta = {}
        for batch in _in_batches(list(only_chunk_ids)):
            names = {f"c{i}": cid for i, cid in enumerate(batch)}
            meta.update(fetch("WHERE chunk_id IN (" + ", ".join(f":{n}" for n in names) + ")", names))
    for cid in img.chunk_ids:
        headings, body, md = meta.get(cid, ("", "", {}))
        img.docs.append(chunk_text(headings, body, md))
        img.metadata.append(md)
    return img


def read_query_adapter(conn) -> np.ndarray | None:
    import sqlalchemy as sa

    try:
        row = conn.execute(sa.text("SELECT metadata FROM index_metadata WHERE id = 'default'")).first()
    except Exception:  # noqa: BLE001 - a store without the table: no adapter
        return None
    if not row or row[0] is None:
        return None
    md = pickle.loads(bytes(row[0]), fix_imports=False) if isinstance(row[0], (bytes, bytearray, memoryview)) else row[0]  # noqa: S301
    adapter = md.get("query_adapter") if isinstance(md, dict) else None
    return None if adapter is None else np.asarray(adapter, dtype=np.float32)


def list_embedded_chunk_ids(conn) -> list[str]:
    import sqlalchemy as sa

    return [r[0] for r in conn.execute(sa.text("SELECT DISTINCT chunk_id FROM chunk_embedding"))]
