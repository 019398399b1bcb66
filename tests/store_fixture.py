"""A RAGLite database for the store tests: the column layout of the reference's tables, nothing else.

Category-b test fixture: only the SCHEMA is taken from /root/reference/src/raglite/_database.py -- `chunk` (:207-224:
id, document_id, index, headings, body, metadata JSON), `chunk_embedding` (:403-430: id auto-increment, chunk_id ->
chunk.id, embedding), `index_metadata` (:434-447: id, version, metadata pickled) -- with the column types the reference
uses on a generic dialect (`NumpyArray` = np.save bytes in a LargeBinary, `PickledObject`, JSON;
/root/reference/src/raglite/_typing.py:57-97).  Rows are written the way `insert_documents` writes them
(/root/reference/src/raglite/_insert.py:114-123,247-272: one `chunk_embedding` row per chunklet vector, in order).
"""

import datetime
import io
import json
import pickle

import numpy as np
import sqlalchemy as sa


def create_store(url: str = "sqlite://") -> sa.engine.Engine:
    engine = sa.create_engine(url)
    md = sa.MetaData()
    sa.Table("document", md, sa.Column("id", sa.String, primary_key=True), sa.Column("filename", sa.String),
             sa.Column("url", sa.String), sa.Column("metadata", sa.JSON))
    sa.Table("chunk", md, sa.Column("id", sa.String, primary_key=True),
             sa.Column("document_id", sa.String, sa.ForeignKey("document.id"), index=True), sa.Column("index", sa.Integer, index=True),
             sa.Column("headings", sa.String), sa.Column("body", sa.String), sa.Column("metadata", sa.JSON))
    sa.Table("chunk_embedding", md, sa.Column("id", sa.Integer, primary_key=True, autoincrement=True),
             sa.Column("chunk_id", sa.String, sa.ForeignKey("chunk.id"), index=True), sa.Column("embedding", sa.LargeBinary))
    sa.Table("index_metadata", md, sa.Column("id", sa.String, primary_key=True), sa.Column("version", sa.DateTime),
             sa.Column("metadata", sa.LargeBinary))
    md.create_all(engine)
    return engine


def _npy(vec: np.ndarray) -> bytes:
    buf = io.BytesIO()
    np.save(buf, vec, allow_pickle=False)  # NumpyArray.process_bind_param
    return buf.getvalue()


def insert_document(engine, doc_id: str, chunks: list[tuple[str, str, str, np.ndarray]], filename: str = "") -> None:
    """chunks: (chunk_id, headings, body, embedding matrix (rows, dim) float16 -- what `_embed.py:140` produces)."""
    with engine.begin() as conn:
        conn.execute(sa.text("INSERT INTO document (id, filename, url, metadata) VALUES (:i, :f, NULL, :m)"),
                     {"i": doc_id, "f": filename or doc_id, "m": json.dumps({})})
        for index, (cid, headings, body, mat) in enumerate(chunks):
            conn.execute(sa.text('INSERT INTO chunk (id, document_id, "index", headings, body, metadata) VALUES (:i, :d, :n, :h, :b, :m)'),
                         {"i": cid, "d": doc_id, "n": index, "h": headings, "b": body,
                          "m": json.dumps({"filename": [filename or doc_id], "topic": [f"t{index % 3}"]})})
            for row in mat:
                conn.execute(sa.text("INSERT INTO chunk_embedding (chunk_id, embedding) VALUES (:c, :e)"), {"c": cid, "e": _npy(row)})


def delete_document(engine, doc_id: str) -> None:
    """`delete_documents` (/root/reference/src/raglite/_delete.py:148-176): embeddings, chunks, document."""
    with engine.begin() as conn:
        ids = [r[0] for r in conn.execute(sa.text("SELECT id FROM chunk WHERE document_id = :d"), {"d": doc_id})]
        for cid in ids:
            conn.execute(sa.text("DELETE FROM chunk_embedding WHERE chunk_id = :c"), {"c": cid})
        conn.execute(sa.text("DELETE FROM chunk WHERE document_id = :d"), {"d": doc_id})
        conn.execute(sa.text("DELETE FROM document WHERE id = :d"), {"d": doc_id})


def set_query_adapter(engine, A: np.ndarray) -> None:
    blob = pickle.dumps({"query_adapter": A}, protocol=pickle.HIGHEST_PROTOCOL, fix_imports=False)  # PickledObject
    with engine.begin() as conn:
        conn.execute(sa.text("DELETE FROM index_metadata WHERE id = 'default'"))
        conn.execute(sa.text("INSERT INTO index_metadata (id, version, metadata) VALUES ('default', :v, :m)"),
                     {"v": datetime.datetime.now(datetime.timezone.utc), "m": blob})


def synthetic_documents(rng, n_docs: int, dim: int, prefix: str = "doc"):
    """n_docs documents of 1..6 chunks of 1..7 unit-norm fp16 rows; chunk ids are 16-hex strings like the reference's."""
    docs = []
    for d in range(n_docs):
        chunks = []
        for c in range(int(rng.integers(1, 7))):
            mat = rng.standard_normal((int(rng.integers(1, 8)), dim))
            mat = (mat / np.linalg.norm(mat, axis=1, keepdims=True)).astype(np.float16)
            cid = f"{rng.integers(0, 2**63):016x}"
            chunks.append((cid, f"# {prefix} {d}", f"body of chunk {c} of {prefix} {d}", mat))
        docs.append((f"{prefix}-{d}", chunks))
    return docs
