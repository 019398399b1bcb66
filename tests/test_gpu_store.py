"""SURVEY.md section 8f-1 for real: the device index built from a RAGLite database, kept in sync with it, and compacted.

The database is SQLite with the reference's table layout (tests/store_fixture.py; rows as
/root/reference/src/raglite/_insert.py:247-272 writes them, deletions as /root/reference/src/raglite/_delete.py:148-176).
Bar: after any sequence of inserts / deletes + sync() (+ compact()) every search returns what a FRESH index built from the
surviving rows returns -- same chunk ids, same scores, bit for bit."""

import numpy as np
import pytest

import raglite_amd
from oracle import oracle
from tests import store_fixture as sf
from tests.util import ragged_offsets

pytestmark = pytest.mark.gpu


def _same_results(a: raglite_amd.GpuIndex, b: raglite_amd.GpuIndex, queries, Qmv) -> None:
    cfg = raglite_amd.HotPathConfig()
    for q in queries:
        ia, sa_ = raglite_amd.vector_search(q, num_results=8, config=cfg, index=a)
        ib, sb = raglite_amd.vector_search(q, num_results=8, config=cfg, index=b)
        assert ia == ib and sa_ == sb
        fa = raglite_amd.vector_search(q, num_results=5, metadata_filter={"topic": "t1"}, config=cfg, index=a)
        fb = raglite_amd.vector_search(q, num_results=5, metadata_filter={"topic": "t1"}, config=cfg, index=b)
        assert fa == fb
    k = min(20, a.index.live()[1])
    sa2, ca = a.index.maxsim_topk_batch(Qmv, k)
    sb2, cb = b.index.maxsim_topk_batch(Qmv, k)
    assert np.array_equal(sa2, sb2)
    assert [[a.chunk_ids[c] for c in row] for row in ca] == [[b.chunk_ids[c] for c in row] for row in cb]


def test_index_from_store_sync_and_compact():
    rng = np.random.default_rng(12)
    dim = 64
    engine = sf.create_store()
    docs = sf.synthetic_documents(rng, 40, dim)
    for doc_id, chunks in docs:
        sf.insert_document(engine, doc_id, chunks, filename=f"{doc_id}.md")
    A = np.linalg.qr(rng.standard_normal((dim, dim)))[0].astype(np.float32)
    sf.set_query_adapter(engine, A)
    gi = raglite_amd.GpuIndex.from_store(engine, metric="cosine")
    n_chunks = sum(len(c) for _, c in docs)
    assert len(gi.chunk_ids) == n_chunks and gi.chunk_ids == sorted(gi.chunk_ids)
    np.testing.assert_array_equal(gi.query_adapter, A)
    # against the hand-assembled index (what round 1 required the caller to build), bit for bit
    by_id = {cid: (h, b, m) for _, chunks in docs for cid, h, b, m in chunks}
    ids = sorted(by_id)
    hand = raglite_amd.GpuIndex(ids, [by_id[i][2] for i in ids], metric="cosine", query_adapter=A,
                                docs=[gi.docs[gi.ordinal_of(i)] for i in ids],
                                metadata=[{"filename": ["x"], "topic": gi.metadata[gi.ordinal_of(i)]["topic"]} for i in ids])
    queries = [rng.standard_normal(dim).astype(np.float16) for _ in range(4)]
    Qmv = rng.standard_normal((5, 12, dim)).astype(np.float32)
    _same_results(gi, hand, queries, Qmv)
    hand.close()
    # the reranker plugin finds chunks by their str(chunk) text (src/raglite/_search.py:394-396)
    ranker = raglite_amd.MaxSimRanker(gi, lambda s: rng.standard_normal((4, dim)).astype(np.float32))
    some = [gi.docs[3], gi.docs[10], gi.docs[7]]
    assert sorted(r.doc_id for r in ranker.rank(query="anything", docs=some).results) == [0, 1, 2]

    # ---- the store changes: 12 documents deleted, 15 inserted; sync() follows it without a rebuild -------------------
    for doc_id, _ in docs[5:17]:
        sf.delete_document(engine, doc_id)
    new_docs = sf.synthetic_documents(rng, 15, dim, prefix="new")
    for doc_id, chunks in new_docs:
        sf.insert_document(engine, doc_id, chunks, filename=f"{doc_id}.md")
    appended, deleted = gi.sync(compact_above=1.0)  # no compaction yet: tombstones stay
    assert appended == sum(len(c) for _, c in new_docs) and deleted == sum(len(c) for _, c in docs[5:17])
    assert gi.index.n_chunks == n_chunks + appended and gi.index.live()[1] == n_chunks + appended - deleted
    fresh = raglite_amd.GpuIndex.from_store(engine, metric="cosine")
    assert sorted(gi._id_to_ordinal) == fresh.chunk_ids  # noqa: SLF001
    _same_results(gi, fresh, queries, Qmv)
    assert gi.sync(compact_above=1.0) == (0, 0)  # idempotent
    # ---- compaction: same answers, no dead rows left, ordinals renumbered consistently ----------------------------------
    rows_before = gi.index.n_rows
    gi.compact()
    assert gi.index.n_chunks == len(gi.chunk_ids) == len(fresh.chunk_ids) and gi.index.n_rows < rows_before
    assert gi.index.live() == (gi.index.n_rows, gi.index.n_chunks)
    assert all(gi.chunk_ids[o] == cid for cid, o in gi._id_to_ordinal.items())  # noqa: SLF001
    _same_results(gi, fresh, queries, Qmv)
    # and it keeps working as a live index afterwards: sync() with automatic compaction
    for doc_id, _ in docs[20:38]:
        sf.delete_document(engine, doc_id)
    gi.sync(compact_above=0.2)
    assert gi.index.live() == (gi.index.n_rows, gi.index.n_chunks)  # compacted by the threshold
    fresh2 = raglite_amd.GpuIndex.from_store(engine, metric="cosine")
    _same_results(gi, fresh2, queries, Qmv)
    for i in (gi, fresh, fresh2):
        i.close()
    with pytest.raises(ValueError, match="insert_documents"):
        raglite_amd.GpuIndex.from_store(sf.create_store())


@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("metric", ["cosine", "dot", "l2"])
def test_rl_index_compact_equals_fresh_index(metric, storage):
    """C-ABI level: tombstone a third of the chunks (incl. the first and the last), compact, and every entry point agrees
    bit for bit with an index created from the surviving rows -- integer data, so ties exercise the renumbering too."""
    rng = np.random.default_rng(3)
    n, dim = 5000, 128
    off = ragged_offsets(rng, n, 1, 12)
    n_chunks = len(off) - 1
    E = oracle.synth_matrix(60, n, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric=metric, storage=storage)
    dead = np.unique(np.concatenate(([0, n_chunks - 1], rng.choice(n_chunks, size=n_chunks // 3, replace=False))))
    idx.delete_chunks(dead)
    remap = idx.compact()
    keep = np.setdiff1d(np.arange(n_chunks), dead)
    assert np.array_equal(np.nonzero(remap >= 0)[0], keep) and np.array_equal(remap[keep], np.arange(len(keep)))
    rows = np.concatenate([np.arange(off[c], off[c + 1]) for c in keep])
    off2 = np.concatenate(([0], np.cumsum(np.diff(off)[keep]))).astype(np.int64)
    assert idx.n_rows == len(rows) and idx.n_chunks == len(keep) and np.array_equal(idx.chunk_offsets, off2)
    fresh = raglite_amd.DeviceIndex(E[rows], off2, metric=metric, storage=storage)
    assert idx.arithmetic == fresh.arithmetic
    Q = oracle.synth_matrix(61, 9, dim, "small_int")
    for a, b in zip(idx.search_rows(Q, 40), fresh.search_rows(Q, 40)):
        assert np.array_equal(a, b)
    for a, b in zip(idx.search_chunks(Q, 60, 15), fresh.search_chunks(Q, 60, 15)):
        assert np.array_equal(a, b)
    Qmv = np.stack([oracle.synth_matrix(62 + i, 20, dim, "small_int") for i in range(4)])
    for a, b in zip(idx.maxsim_topk_batch(Qmv, 30), fresh.maxsim_topk_batch(Qmv, 30)):
        assert np.array_equal(a, b)
    ws, wc = oracle.maxsim_topk(E[rows], off2, Qmv[0], 30, np.float32)
    gs, gc = idx.maxsim_topk(Qmv[0], 30)
    assert np.array_equal(gc, wc) and np.array_equal(gs, ws)
    # life goes on: append after compaction, delete again, compact again
    idx.append(E[:50], np.full(10, 5))
    fresh.append(E[:50], np.full(10, 5))
    idx.delete_chunks([1, 2])
    fresh.delete_chunks([1, 2])
    assert np.array_equal(idx.compact(), fresh.compact())
    for a, b in zip(idx.search_chunks(Q, 60, 15), fresh.search_chunks(Q, 60, 15)):
        assert np.array_equal(a, b)
    assert np.array_equal(idx.compact(), np.arange(idx.n_chunks))  # nothing to do: identity
    idx.close()
    fresh.close()
