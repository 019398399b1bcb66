"""Host-side logic of the mirror that needs no GPU (fake device index)."""

import numpy as np
import pytest

import raglite_amd
from raglite_amd import _search
from raglite_amd._sharded import group_chunk_max_host, merge_topk_host, shard_bounds_by_chunk
from oracle import oracle


class _FakeDeviceIndex:
    """Stands in for raglite_amd.DeviceIndex: records the call, answers with the oracle."""

    def __init__(self, E, off, metric):
        self.E, self.off, self.metric = E, off, metric
        self.n_rows, self.n_chunks = len(E), len(off) - 1
        self.chunk_offsets = np.asarray(off, dtype=np.int64)
        self.calls = []
        self.alive = np.ones(self.n_chunks, bool)

    def append(self, rows, sizes):
        self.E = np.concatenate([self.E, rows])
        self.off = np.concatenate([self.off, self.off[-1] + np.cumsum(sizes)])
        self.alive = np.concatenate([self.alive, np.ones(len(sizes), bool)])
        self.n_rows, self.n_chunks = len(self.E), len(self.off) - 1
        self.chunk_offsets = np.asarray(self.off, dtype=np.int64)

    def delete_chunks(self, ords):
        self.alive[np.asarray(ords)] = False

    def search_chunks(self, q, num_hits, k, chunk_filter=None, rank_limit=None):
        if np.ndim(q) == 2:  # batched form: one row per query
            outs = [self.search_chunks(qq, num_hits, k, chunk_filter, rank_limit) for qq in q]
            return tuple(np.stack([o[j] for o in outs]) for j in range(3))
        self.calls.append((num_hits, k))
        self.rank_limits = getattr(self, "rank_limits", []) + [rank_limit]
        r2c = np.repeat(np.arange(self.n_chunks), np.diff(self.off))
        ok = np.ones(self.n_chunks, bool) if chunk_filter is None else np.asarray(chunk_filter, bool)
        if rank_limit:
            s, c = oracle.search_chunks_ranked(self.E, r2c, q, num_hits, k, ok, rank_limit, self.alive, self.metric, np.float32)
        else:
            s, c = oracle.search_chunks_filtered(self.E, r2c, q, num_hits, k, ok & self.alive, self.metric, np.float32)
        out_s = np.full(k, -np.inf, np.float32); out_c = np.full(k, -1, np.int32)
        out_s[: len(s)] = s; out_c[: len(c)] = c
        return out_s, out_c, np.int32(len(c))

    def chunk_best_rows(self, Q, cand):
        out = np.full(cand.shape, -1, np.int32)
        for b in range(len(Q)):
            for j, c in enumerate(cand[b]):
                if c >= 0:
                    out[b, j] = self.off[c] + oracle.best_row(self.E[self.off[c] : self.off[c + 1]], Q[b])
        return out

    def gather_rows(self, rows):
        return self.E[np.asarray(rows)].astype(np.float32)

    def maxsim_rerank(self, qv, cand):
        return np.stack([oracle.maxsim_candidates(self.E, self.off, qv[i], cand[i], np.float32)
                         for i in range(len(qv))]).astype(np.float32)


def _gpu_index(n_chunks=30, dim=16, seed=0, **kw):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, 6, size=n_chunks)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    E = rng.standard_normal((off[-1], dim)).astype(np.float32)
    gi = _search.GpuIndex.__new__(_search.GpuIndex)
    gi.chunk_ids = [f"chunk{i:04d}" for i in range(n_chunks)]
    gi.index = _FakeDeviceIndex(E, off, kw.get("metric", "cosine"))
    gi.metric = kw.get("metric", "cosine")
    gi.query_adapter = kw.get("query_adapter")
    gi.docs = [f"doc text {i}" for i in range(n_chunks)]
    gi._doc_to_ordinal = {d: i for i, d in enumerate(gi.docs)}
    gi.metadata = None
    gi._id_to_ordinal = {c: i for i, c in enumerate(gi.chunk_ids)}
    return gi


def test_vector_search_contract_and_num_hits():
    """`tests/test_search.py:36-60` of the reference: list[str] ids, list[float] scores, same length."""
    gi = _gpu_index()
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    q = np.random.default_rng(1).standard_normal(16).astype(np.float16)
    ids, scores = raglite_amd.vector_search(q, num_results=5, config=cfg, index=gi)
    assert len(ids) == len(scores) == 5
    assert all(isinstance(i, str) for i in ids) and all(isinstance(s, float) for s in scores)
    assert gi.index.calls[-1] == (40, 5)  # round(4 * 2048 / 2048) * max(5, 10)
    ids, _ = raglite_amd.vector_search(q, num_results=40, oversample=4, config=cfg, index=gi)
    assert gi.index.calls[-1] == (160, 40)
    cfg2 = raglite_amd.HotPathConfig(chunk_max_size=1024, vector_search_query_adapter=False)
    raglite_amd.vector_search(q, num_results=8, config=cfg2, index=gi)
    assert gi.index.calls[-1] == (20, 8)
    assert scores == sorted(scores, reverse=True)


def test_metadata_filter_is_pushed_down_as_chunk_mask():
    """`_search.py:84-119`: JSON containment evaluated on the host metadata, ranking restricted to matching chunks."""
    gi = _gpu_index(n_chunks=40)
    gi.metadata = [{"topic": ["a", "b"] if i % 3 == 0 else ["c"], "year": 2020 + i % 2} for i in range(40)]
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    q = np.random.default_rng(3).standard_normal(16).astype(np.float32)
    ids, scores = raglite_amd.vector_search(q, num_results=6, metadata_filter={"topic": "a", "year": 2020},
                                            config=cfg, index=gi)
    want = {f"chunk{i:04d}" for i in range(40) if i % 3 == 0 and i % 2 == 0}
    assert ids and set(ids) <= want and scores == sorted(scores, reverse=True)
    assert raglite_amd.vector_search(q, num_results=6, metadata_filter={"topic": "zzz"}, config=cfg, index=gi) == ([], [])
    gi.metadata = None
    with pytest.raises(ValueError):
        raglite_amd.vector_search(q, num_results=6, metadata_filter={"topic": "a"}, config=cfg, index=gi)


def test_insert_and_delete_chunks_keep_ordinals_stable():
    """`insert_documents` / `delete_documents` on the host mirror (`_insert.py:247-272`, `_delete.py:148-176`)."""
    gi = _gpu_index(n_chunks=12)
    gi.metadata = [{"k": i} for i in range(12)]
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    rng = np.random.default_rng(5)
    new = [rng.standard_normal((3, 16)).astype(np.float32), rng.standard_normal((1, 16)).astype(np.float32)]
    gi.insert_chunks(["new-a", "new-b"], new, docs=["doc a", "doc b"], metadata=[{"k": 100}, {"k": 101}])
    assert gi.ordinal_of("new-b") == 13 and gi.ordinal_of_doc("doc a") == 12 and len(gi.metadata) == 14
    ids, _ = raglite_amd.vector_search(new[1][0], num_results=1, config=cfg, index=gi)
    assert ids == ["new-b"]
    with pytest.raises(ValueError):
        gi.insert_chunks(["new-a"], [new[0]], docs=["x"], metadata=[{}])
    assert gi.delete_chunks(["new-b", "chunk0003", "not-there"]) == 2
    ids, _ = raglite_amd.vector_search(new[1][0], num_results=14, config=cfg, index=gi)
    assert "new-b" not in ids and "chunk0003" not in ids and len(ids) == 12
    assert gi.delete_chunks(["new-b"]) == 0


def test_vector_search_empty_index_returns_empty_lists():
    gi = _gpu_index()
    gi.index.n_rows = 0
    assert raglite_amd.vector_search(np.zeros(16, np.float16), num_results=5, index=gi) == ([], [])


def test_search_method_plugin_signature():
    gi = _gpu_index()
    method = raglite_amd.GpuVectorSearch(gi)
    cfg = raglite_amd.HotPathConfig(search_method=method, vector_search_query_adapter=False)
    ids, scores = cfg.search_method(np.ones(16, np.float16), num_results=3, metadata_filter=None, config=cfg)
    assert len(ids) == 3 and len(scores) == 3


def test_rerank_chunks_identity_without_reranker():
    """`src/raglite/_search.py:376-377` / `tests/test_rerank.py:64-70`: identity when reranker is None."""
    cfg = raglite_amd.HotPathConfig(reranker=None)
    chunks = ["c", "a", "b"]
    assert raglite_amd.rerank_chunks("q", chunks, config=cfg, chunk_lookup=lambda ids: list(ids)) == chunks
    assert raglite_amd.rerank_chunks("q", [], config=cfg) == []


def test_maxsim_ranker_orders_by_score_and_plugs_into_rerank_chunks():
    gi = _gpu_index(n_chunks=12, dim=8, seed=3)
    rng = np.random.default_rng(5)
    qv = rng.standard_normal((4, 8)).astype(np.float32)
    ranker = raglite_amd.MaxSimRanker(gi, lambda query: qv)
    docs = [gi.docs[i] for i in (7, 2, 9, 0, 5)]
    res = ranker.rank(query="anything", docs=docs)
    exp = oracle.maxsim_candidates(gi.index.E, gi.index.off, qv, [7, 2, 9, 0, 5], np.float32)
    order = np.argsort(-exp, kind="stable")
    assert [r.doc_id for r in res.results] == order.tolist()
    assert [r.rank for r in res.results] == [1, 2, 3, 4, 5]

    class _Chunk:  # what the reference hands over: objects whose str() is the chunk text
        def __init__(self, text): self.text = text
        def __str__(self): return self.text

    cfg = raglite_amd.HotPathConfig(reranker=ranker)
    chunks = [_Chunk(d) for d in docs]
    out = raglite_amd.rerank_chunks("anything", chunks, config=cfg)
    assert [c.text for c in out] == [docs[i] for i in order]
    cfg_dict = raglite_amd.HotPathConfig(reranker={"en": None, "other": ranker})
    out2 = raglite_amd.rerank_chunks("anything", chunks, config=cfg_dict)
    assert [c.text for c in out2] == [c.text for c in out]


def test_merge_topk_host_matches_oracle():
    rng = np.random.default_rng(7)
    world, B, k = 4, 5, 10
    scores = rng.integers(-5, 6, size=(world, B, k)).astype(np.float32)
    scores = -np.sort(-scores, axis=2)
    ids = np.stack([rng.permutation(1000)[: B * k].reshape(B, k) + w * 1000 for w in range(world)]).astype(np.int64)
    ids[3, :, -2:] = -1; scores[3, :, -2:] = -np.inf  # padding from a short shard
    ms, mi = merge_topk_host(scores, ids, k)
    for b in range(B):
        valid = [ids[w, b][ids[w, b] >= 0] for w in range(world)]
        vs = [scores[w, b][ids[w, b] >= 0] for w in range(world)]
        es, ei = oracle.merge_topk(vs, valid, k)
        assert np.array_equal(mi[b], ei) and np.array_equal(ms[b], es)


def test_shard_bounds_match_oracle():
    rng = np.random.default_rng(9)
    for _ in range(50):
        sizes = rng.integers(0, 20, size=int(rng.integers(1, 60)))
        off = np.concatenate(([0], np.cumsum(sizes)))
        for w in (1, 2, 3, 8):
            assert shard_bounds_by_chunk(off, w) == oracle.shard_bounds_by_chunk(off, w)


def test_group_chunk_max_host_matches_oracle():
    """Hits arrive sorted by (score desc, row asc) and chunk ordinals grow with row ids, as on the GPU."""
    rng = np.random.default_rng(11)
    for _ in range(20):
        n_rows = 200
        r2c = np.sort(rng.integers(0, 30, size=n_rows))
        rows = rng.permutation(n_rows)[:40]
        sc = rng.integers(0, 9, size=40).astype(np.float32)  # many ties
        order = np.lexsort((rows, -sc))
        rows, sc = rows[order], sc[order]
        out_s, out_c, n = group_chunk_max_host(sc[None], r2c[rows][None], 5)
        es, ec = oracle.group_chunk_max(sc, r2c[rows], 5)
        assert out_c[0, : n[0]].tolist() == ec.tolist()
        np.testing.assert_array_equal(out_s[0, : n[0]], es)


def test_torch_token_embedder_surface_on_cpu():
    """the PyTorch token-level embedder exposes exactly what `_embed.py` touches; a tiny shape runs
    on the CPU here (the encoder is plain PyTorch), the pooling seam is exercised on the GPU in test_gpu_parity."""
    import torch

    from raglite_amd import _embed
    from raglite_amd._torch_embedder import EncoderShape, HashTokenizer, TorchTokenEmbedder

    shape = EncoderShape(vocab_size=5000, hidden=32, layers=2, heads=4, ffn=64, max_positions=300, n_ctx=256)
    emb = TorchTokenEmbedder(shape, device="cpu", seed=3)
    tok = HashTokenizer(5000)
    text = "Some text, with punctuation; and numbers 12345678."
    assert tok.decode(tok.encode(text)) == text
    assert emb.detokenize(emb.tokenize(text.encode(), add_bos=True)) == text.encode()
    assert _embed._sentinel_token_ids(emb)  # the sentinel character is a token of its own
    sents = ["Hello world this is one sentence. ", "And another, shorter one. ", "Third! "]
    counts = _embed.count_sentence_tokens(sents, emb)
    assert len(counts) == 3 and counts.sum() > 0
    whole = emb.embed("".join(sents))
    assert whole.shape == (len(tok.encode("".join(sents))) + 2, 32) and whole.dtype == torch.float32
    batch = emb.embed(sents)
    for s, m in zip(sents, batch):  # padding + attention mask: a string's rows do not depend on its batch mates
        torch.testing.assert_close(m, emb.embed(s), atol=1e-5, rtol=0)
    again = TorchTokenEmbedder(shape, device="cpu", seed=3).embed(sents[0])
    torch.testing.assert_close(again, batch[0], atol=1e-5, rtol=0)  # seeded init
    plan = _embed.plan_segments(counts, emb.n_ctx(), emb.n_batch)
    assert plan == [(0, 0, 3)]
    assert _embed.split_rows(len(whole), counts).sum() == len(whole)


@pytest.mark.parametrize("metric", ["cosine", "dot"])
def test_update_query_adapter_matches_oracle_loop(metric):
    """the batched host mirror (one search, one best-row call, one gather) computes the same adapter
    as the reference's per-eval loop restated in oracle.update_query_adapter (whose arithmetic is pinned against the
    reference's own lines by tests/golden/query_adapter.npz)."""
    gi = _gpu_index(n_chunks=60, dim=16, seed=4, metric=metric)
    gi.index.E = (gi.index.E / np.linalg.norm(gi.index.E, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    rng = np.random.default_rng(6)
    evals = []
    for _ in range(25):
        target = int(rng.integers(0, 60))
        row = gi.index.E[gi.index.off[target]]
        q = (row + 0.6 * rng.standard_normal(16)).astype(np.float16)
        evals.append((q, [gi.chunk_ids[target], gi.chunk_ids[(target + 7) % 60]]))
    evals.append((rng.standard_normal(16).astype(np.float16), ["no-such-chunk"]))  # retrieves nothing relevant: skipped
    cfg = raglite_amd.HotPathConfig(vector_search_distance_metric=metric)
    A = raglite_amd.update_query_adapter(evals, optimize_top_k=8, config=cfg, index=gi)
    want, Q, T = oracle.update_query_adapter(evals, gi.index.E, gi.index.off, gi.chunk_ids, optimize_top_k=8,
                                             metric=metric, dtype=np.float32)
    assert len(Q) < len(evals)  # at least the last eval was skipped
    np.testing.assert_allclose(A, want, rtol=0, atol=1e-9)
    np.testing.assert_allclose(gi.query_adapter, want.astype(np.float32), atol=1e-6)
    if metric == "cosine":
        np.testing.assert_allclose(A @ A.T, np.eye(16), atol=1e-9)
    with pytest.raises(ValueError):
        raglite_amd.update_query_adapter([], config=cfg, index=gi)
    with pytest.raises(ValueError):
        raglite_amd.update_query_adapter(evals, config=raglite_amd.HotPathConfig(vector_search_distance_metric="l2"), index=gi)
    with pytest.raises(ValueError):
        raglite_amd.update_query_adapter([(evals[0][0], ["no-such-chunk"])], config=cfg, index=gi)


def test_hybrid_search_fuses_vector_and_keyword_rankings():
    """`_search.py:255-279`: oversampled vector + keyword rankings fused by RRF with weights 0.75 / 0.25."""
    gi = _gpu_index(n_chunks=30)
    cfg = raglite_amd.HotPathConfig(vector_search_query_adapter=False)
    q = np.random.default_rng(8).standard_normal(16).astype(np.float32)
    vs, _ = raglite_amd.vector_search(q, num_results=6, config=cfg, index=gi)
    calls = []

    def keyword_search(query, *, num_results, metadata_filter=None, config=None):
        calls.append(num_results)
        return [vs[3], "chunk0029", vs[0]], [3.0, 2.0, 1.0]

    ids, scores = raglite_amd.hybrid_search(q, num_results=3, config=cfg, index=gi, keyword_search=keyword_search)
    want_ids, want_scores = raglite_amd.reciprocal_rank_fusion([vs, [vs[3], "chunk0029", vs[0]]], weights=[0.75, 0.25])
    assert calls == [6] and ids == want_ids[:3] and scores == want_scores[:3]
    only_vs, _ = raglite_amd.hybrid_search(q, num_results=3, config=cfg, index=gi)
    assert only_vs == vs[:3]


def _hf_pad(rows, pad):
    import torch

    T = max(len(r) for r in rows)  # noqa: N806
    ids = torch.full((len(rows), T), pad, dtype=torch.long)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = torch.tensor(r)
    lengths = torch.tensor([len(r) for r in rows])
    return ids, lengths, (torch.arange(T)[None, :] < lengths[:, None]).long()


def test_token_encoder_matches_huggingface_xlm_roberta():
    """SURVEY.md 8f-2 pin: bge-m3 is an XLM-RoBERTa encoder; with the same (random) weights, loaded through
    `load_hf_state_dict`, the token-level outputs equal Hugging Face's `XLMRobertaModel` -- embeddings incl. the
    segment-type vector and the pad-offset positions, post-LN layers, padding mask."""
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaModel

    from raglite_amd._torch_embedder import EncoderShape, TorchTokenEmbedder

    torch.manual_seed(11)
    cfg = XLMRobertaConfig(vocab_size=900, hidden_size=48, num_hidden_layers=3, num_attention_heads=4, intermediate_size=96,
                           max_position_embeddings=70, type_vocab_size=1, layer_norm_eps=1e-5, hidden_dropout_prob=0.0,
                           attention_probs_dropout_prob=0.0, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    hf = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in hf.parameters():  # HF initialises biases / LayerNorm to 0 / 1: make every parameter count
            p.add_(0.05 * torch.randn_like(p))
    shape = EncoderShape(vocab_size=900, hidden=48, layers=3, heads=4, ffn=96, max_positions=70, n_ctx=64)
    emb = TorchTokenEmbedder(shape, device="cpu", dtype=torch.float32)
    emb.load_hf_state_dict(hf.state_dict())
    g = torch.Generator().manual_seed(5)
    rows = [[0, *torch.randint(3, 900, (n,), generator=g).tolist(), 2] for n in (17, 5, 62, 1)]
    ids, lengths, mask = _hf_pad(rows, 1)
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        got = emb.encoder(ids, lengths)
    for i, r in enumerate(rows):
        torch.testing.assert_close(got[i, : len(r)], want[i, : len(r)], atol=2e-5, rtol=1e-5)


def test_cross_encoder_matches_huggingface_bert_classifier():
    """the cross-encoder `BaseRanker` (FlashRank's ms-marco MiniLM is a BERT sequence classifier).
    With the same weights the pair logits equal Hugging Face's `BertForSequenceClassification`; scores are the
    sigmoid; `rank` orders best-first with the reference's `doc_id` plumbing (`_search.py:394-396`)."""
    import torch
    from transformers import BertConfig, BertForSequenceClassification

    import raglite_amd
    from raglite_amd._cross_encoder import CrossEncoderShape, TorchCrossEncoderRanker, truncate_pair

    torch.manual_seed(12)
    cfg = BertConfig(vocab_size=1200, hidden_size=48, num_hidden_layers=2, num_attention_heads=4, intermediate_size=96,
                     max_position_embeddings=64, type_vocab_size=2, num_labels=1, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, layer_norm_eps=1e-12)
    hf = BertForSequenceClassification(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    shape = CrossEncoderShape(vocab_size=1200, hidden=48, layers=2, heads=4, ffn=96, max_positions=64, n_ctx=64)
    rk = TorchCrossEncoderRanker(shape, device="cpu", dtype=torch.float32, pairs_per_batch=3)
    rk.load_hf_state_dict(hf.state_dict())
    query = "what is late chunking"
    docs = ["late chunking pools token embeddings per sentence. " * 4, "short", "", "an unrelated passage about ducks, geese and swans " * 2,
            "late chunking", "x y z " * 40]
    pairs = rk.encode_pairs(query, docs)
    assert max(len(p[0]) for p in pairs) == 64 and pairs[2][0][-2:] == [102, 102]  # truncated to max_length; empty passage
    ids, lengths, mask = _hf_pad([p[0] for p in pairs], 0)
    first = torch.tensor([p[1] for p in pairs])
    ar = torch.arange(ids.shape[1])
    types = ((ar[None, :] >= first[:, None]) & (ar[None, :] < lengths[:, None])).long()
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=mask, token_type_ids=types).logits[:, 0]
    got = rk.logits(query, docs)
    torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(rk.score(query, docs), torch.sigmoid(want).numpy(), atol=1e-6)
    res = rk.rank(query=query, docs=docs)
    order = np.lexsort((np.arange(len(docs)), -torch.sigmoid(want).numpy()))
    assert [r.doc_id for r in res.results] == order.tolist() and res.results[0].rank == 1
    assert rk.rank(query=query, docs=[]).results == []

    # the reranker plugs into rerank_chunks exactly like the reference's (`_search.py:364-397`)
    class _Chunk:
        def __init__(self, t): self.t = t
        def __str__(self): return self.t

    chunks = [_Chunk(d) for d in docs]
    out = raglite_amd.rerank_chunks(query, chunks, config=raglite_amd.HotPathConfig(reranker=rk))
    assert [c.t for c in out] == [docs[i] for i in order]

    # longest_first truncation in closed form == the one-token-at-a-time definition
    for nq in range(0, 40, 3):
        for nd in range(0, 40, 3):
            for budget in (0, 1, 7, 20, 61):
                a, b = nq, nd
                while a + b > budget:
                    if a > b:
                        a -= 1
                    else:
                        b -= 1
                assert truncate_pair(nq, nd, budget) == (a, b), (nq, nd, budget)


def test_update_query_adapter_embeds_questions_one_by_one_under_late_chunking(monkeypatch):
    """A late-chunking embedder pools a LIST of strings as the sentences of one document, so the questions of the
    evals must be embedded one call each, as the reference does (`/root/reference/src/raglite/_query_adapter.py:160`:
    `embed_strings([eval_.question])[0]`) and as `vector_search` does at query time; a standard embedder is batched."""
    from raglite_amd import _query_adapter

    gi = _gpu_index(n_chunks=40, dim=16, seed=9, metric="cosine")
    gi.index.E = (gi.index.E / np.linalg.norm(gi.index.E, axis=1, keepdims=True)).astype(np.float16).astype(np.float32)
    rng = np.random.default_rng(3)
    targets = [int(rng.integers(0, 40)) for _ in range(12)]
    vec_of = {f"question {i}": (gi.index.E[gi.index.off[t]] + 0.5 * rng.standard_normal(16)).astype(np.float16)
              for i, t in enumerate(targets)}
    evals = [(text, [gi.chunk_ids[t], gi.chunk_ids[(t + 5) % 40]]) for text, t in zip(vec_of, targets)]
    calls = []

    def fake_embed_strings(strings, *, config=None, embedder=None):
        calls.append(list(strings))
        return np.stack([vec_of[s] for s in strings])

    monkeypatch.setattr(_query_adapter, "embed_strings", fake_embed_strings)
    late = raglite_amd.HotPathConfig(embedder="llama-cpp-python/some/model.gguf@512")
    raglite_amd.update_query_adapter(evals, optimize_top_k=8, config=late, index=gi)
    assert calls == [[text] for text in vec_of]  # one call per question
    calls.clear()
    standard = raglite_amd.HotPathConfig(embedder="text-embedding-3-large")
    raglite_amd.update_query_adapter(evals, optimize_top_k=8, config=standard, index=gi)
    assert calls == [list(vec_of)]  # one batched call


def test_store_reader_orders_rows_by_chunk_then_id_and_decodes_every_dialect():
    """the reader's contract, no GPU -- rows come back ordered by (chunk_id, id) whatever the insertion
    order, `str(chunk)` is rebuilt like /root/reference/src/raglite/_database.py:300-324, metadata is the JSON dict,
    the query adapter is un-pickled, and the three embedding encodings of /root/reference/src/raglite/_typing.py decode."""
    from raglite_amd import _store
    from tests import store_fixture as sf

    rng = np.random.default_rng(0)
    engine = sf.create_store()
    docs = sf.synthetic_documents(rng, 6, 8)
    for doc_id, chunks in docs:
        sf.insert_document(engine, doc_id, chunks, filename=f"{doc_id}.md")
    A = rng.standard_normal((8, 8)).astype(np.float32)
    sf.set_query_adapter(engine, A)
    with engine.connect() as conn:
        img = _store.read_chunks(conn)
        adapter = _store.read_query_adapter(conn)
        ids = _store.list_embedded_chunk_ids(conn)
    by_id = {cid: (h, b, m) for _, chunks in docs for cid, h, b, m in chunks}
    assert img.chunk_ids == sorted(by_id) and set(ids) == set(by_id)
    at = 0
    for cid, size, doc, md in zip(img.chunk_ids, img.sizes, img.docs, img.metadata):
        h, b, m = by_id[cid]
        assert size == len(m)
        np.testing.assert_array_equal(np.vstack(img.rows[at : at + size]), m.astype(np.float32))  # insertion (= id) order
        at += size
        assert doc.startswith("---\nfilename: ") and doc.endswith(f"{h}\n\n{b}") and md["topic"][0].startswith("t")
    np.testing.assert_array_equal(adapter, A)
    only = img.chunk_ids[1:3]
    with engine.connect() as conn:
        part = _store.read_chunks(conn, only)
    assert part.chunk_ids == only and part.docs == img.docs[1:3]
    v = rng.standard_normal(5).astype(np.float16)
    np.testing.assert_array_equal(_store.decode_embedding(v.astype(np.float32).tolist()), v.astype(np.float32))  # DuckDB FLOAT[d]
    np.testing.assert_array_equal(_store.decode_embedding("[" + ",".join(str(x) for x in v) + "]"), v.astype(np.float32))  # halfvec text
    np.testing.assert_array_equal(_store.decode_embedding(sf._npy(v)), v.astype(np.float32))  # NumpyArray bytes
    assert _store._connection(engine)[1] and not _store._connection(engine.connect())[1]


def test_language_aware_reranker_selection():
    """`/root/reference/src/raglite/_search.py:378-392`: a dict of rerankers is resolved by language -- one detected language
    with an entry -> that ranker; mixed languages, an unknown language, a failing detector or no detector -> "other"."""

    class _R:
        def __init__(self, name):
            self.name, self.calls = name, 0

        def rank(self, query, docs):
            self.calls += 1
            return type("Res", (), {"results": [type("X", (), {"doc_id": i})() for i in reversed(range(len(docs)))]})()

    en, nl, other = _R("en"), _R("nl"), _R("other")
    rer = {"en": en, "nl": nl, "other": other}
    lang_of = {"hello world": "en", "good morning": "en", "hallo wereld": "nl", "bonjour": "fr"}
    detect = lang_of.__getitem__
    assert raglite_amd.select_reranker(rer, "hello world", ["good morning"], detect) is en
    assert raglite_amd.select_reranker(rer, "hallo wereld", ["hallo wereld"], detect) is nl
    assert raglite_amd.select_reranker(rer, "hello world", ["hallo wereld"], detect) is other       # mixed
    assert raglite_amd.select_reranker(rer, "bonjour", ["bonjour"], detect) is other                # no entry for fr
    assert raglite_amd.select_reranker(rer, "???", ["hello world"], detect) is other                # detector raised (KeyError)
    assert raglite_amd.select_reranker(en, "x", ["y"], detect) is en                                 # not a dict: as configured
    assert raglite_amd.select_reranker({"en": en}, "bonjour", ["bonjour"], detect) is None           # no "other": no reranking
    cfg = raglite_amd.HotPathConfig(reranker=rer)
    out = raglite_amd.rerank_chunks("hello world", ["good morning", "hello world"], config=cfg, detect=detect,
                                    chunk_lookup=lambda ids: list(ids))
    assert out == ["hello world", "good morning"] and en.calls == 1 and other.calls == 0
    raglite_amd.set_language_detector(detect)  # process-wide default (the reference's module-level langdetect import)
    try:
        raglite_amd.rerank_chunks("hallo wereld", ["hallo wereld"], config=cfg, chunk_lookup=lambda ids: list(ids))
        assert nl.calls == 1
    finally:
        raglite_amd.set_language_detector(None)
    raglite_amd.rerank_chunks("hallo wereld", ["hallo wereld"], config=cfg, chunk_lookup=lambda ids: list(ids))
    assert other.calls == 1  # no langdetect in this image: "other"


def test_vector_search_switches_branch_on_the_matching_row_count(monkeypatch):
    """`_search.py:97-105`: more than 100 000 matching embedding rows -> order first (LIMIT 1 000 000), then filter."""
    from raglite_amd import _search

    rng = np.random.default_rng(3)
    n_chunks, dim = 60, 16
    sizes = rng.integers(1, 4, n_chunks)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    E = rng.standard_normal((int(off[-1]), dim)).astype(np.float32)
    fake = _FakeDeviceIndex(E, off, "cosine")
    gi = raglite_amd.GpuIndex.__new__(raglite_amd.GpuIndex)
    gi.index, gi.chunk_ids, gi.query_adapter = fake, [f"c{i}" for i in range(n_chunks)], None
    gi.metadata = [{"k": "x" if i % 2 else "y"} for i in range(n_chunks)]
    q = rng.standard_normal(dim).astype(np.float32)
    raglite_amd.vector_search(q, num_results=3, metadata_filter={"k": "x"}, index=gi)
    assert fake.rank_limits[-1] is None  # a few dozen matching rows: filter first
    monkeypatch.setattr(_search, "FILTER_FIRST_MAX_ROWS", 5)
    ids, _ = raglite_amd.vector_search(q, num_results=3, metadata_filter={"k": "x"}, index=gi)
    assert fake.rank_limits[-1] == 1_000_000 and len(ids) == 3


def test_oracle_order_first_then_filter_branch():
    """`oracle.search_rows_ranked` (`_search.py:120-141`): hand-checkable case -- the cut to the `rank_limit` nearest rows
    happens BEFORE the filter, so a matching row outside the cut never surfaces, and a covering limit is the filter-first
    result."""
    from oracle import oracle

    E = np.eye(6, dtype=np.float32)[:, :4].copy()
    E[:, 0] = [0.9, 0.8, 0.7, 0.6, 0.5, 0.4]  # dot with q = e0 ranks the rows 0, 1, 2, 3, 4, 5
    q = np.array([1, 0, 0, 0], np.float32)
    r2c = np.array([0, 0, 1, 1, 2, 2])
    ok = np.array([False, True, True])  # chunk 0 fails the filter
    s, r = oracle.search_rows_ranked(E, r2c, q, 3, ok, 3, None, "dot")
    assert r.tolist() == [2, -1, -1] and abs(s[0] - 1.7) < 1e-6 and np.isneginf(s[1:]).all()  # rows 0, 1 cut in, filtered out; 3.. cut out
    s, r = oracle.search_rows_ranked(E, r2c, q, 3, ok, 6, None, "dot")
    fs, fr = oracle.search_rows_filtered(E, r2c, q, 3, ok, "dot")
    assert r.tolist() == fr.tolist() == [2, 3, 4]
    live = np.array([True, False, True])  # chunk 1 deleted: rows 2, 3 are not in the table, the cut takes 0, 1, 4
    s, r = oracle.search_rows_ranked(E, r2c, q, 3, ok, 3, live, "dot")
    assert r.tolist() == [4, -1, -1]
    cs, cc = oracle.search_chunks_ranked(E, r2c, q, 3, 2, ok, 3, live, "dot")
    assert cc.tolist() == [2] and abs(cs[0] - 1.5) < 1e-6


def test_device_side_merge_and_group_by_equal_the_host_code():
    """`merge_order_torch` / `group_chunk_max_torch` (what ShardedIndex.search_chunks runs for CUDA tensors, no host sync) against the
    NumPy host code on lists full of ties, padding, -inf and NaN."""
    import torch

    from raglite_amd import _sharded

    rng = np.random.default_rng(3)
    B, M, k = 7, 60, 9
    s = rng.integers(-3, 4, (B, M)).astype(np.float32)       # heavy ties
    s[rng.random((B, M)) < 0.05] = -np.inf
    s[rng.random((B, M)) < 0.05] = np.nan
    i = np.stack([rng.permutation(1000)[:M] for _ in range(B)]).astype(np.int64)
    i[rng.random((B, M)) < 0.2] = -1                         # padding
    i[3] = -1                                                # a query without any hit
    for n in (1, 25, M, M + 5):
        order, n_valid = _sharded._merge_order(s, i, n)
        t_order, t_valid = _sharded.merge_order_torch(torch.from_numpy(s), torch.from_numpy(i), n)
        assert np.array_equal(n_valid, t_valid.numpy())
        for b in range(B):  # (positions of padding entries beyond n_valid are arbitrary: compare the real part)
            assert np.array_equal(order[b, : n_valid[b]], t_order[b, : n_valid[b]].numpy())
    chunks = rng.integers(0, 12, (B, M)).astype(np.int64)
    chunks[i < 0] = -1
    order, n_valid = _sharded._merge_order(s, i, 40)
    real = np.arange(order.shape[1])[None, :] < n_valid[:, None]
    ms = np.where(real, np.take_along_axis(s, order, axis=1), -np.inf).astype(np.float32)
    mc = np.where(real, np.take_along_axis(chunks, order, axis=1), -1)
    hs, hc, hn = _sharded.group_chunk_max_host(ms, mc, k)
    ts, tc, tn = _sharded.group_chunk_max_torch(torch.from_numpy(ms), torch.from_numpy(mc), k)
    assert np.array_equal(hn, tn.numpy()) and np.array_equal(hc, tc.numpy())
    np.testing.assert_array_equal(hs, ts.numpy())            # (NaN == NaN here: assert_array_equal treats them as equal)


def test_sharded_search_chunks_device_branch_on_cpu_tensors(monkeypatch):
    """The CUDA branch of ShardedIndex.search_chunks (gathers + torch merge / group-by) driven with CPU tensors: world of one,
    must equal the NumPy branch and the oracle."""
    import torch

    from oracle import oracle
    from raglite_amd import _sharded
    from tests.util import ragged_offsets

    rng = np.random.default_rng(4)
    off = ragged_offsets(rng, 300, 1, 7)
    E = oracle.synth_matrix(21, 300, 16, "small_int")
    Q = oracle.synth_matrix(22, 4, 16, "small_int")
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))

    class Local:
        def __init__(self, as_torch):
            self.as_torch = as_torch

        def search_rows(self, q, k):
            q2 = np.atleast_2d(q.numpy() if hasattr(q, "numpy") else q)
            S = np.full((len(q2), k), -np.inf, np.float32)
            I = np.full((len(q2), k), -1, np.int32)
            for b, qq in enumerate(q2):
                s, i = oracle.search_rows(E, qq, k, "dot", np.float64)
                S[b, : len(s)], I[b, : len(i)] = s, i
            return (torch.from_numpy(S), torch.from_numpy(I)) if self.as_torch else (S, I)

    host = _sharded.ShardedIndex(Local(False), row_base=0, chunk_base=0, local_chunk_offsets=off).search_chunks(Q, 30, 5)
    monkeypatch.setattr(_sharded, "_is_cuda", lambda x: hasattr(x, "dim"))
    dev = _sharded.ShardedIndex(Local(True), row_base=0, chunk_base=0, local_chunk_offsets=off).search_chunks(torch.from_numpy(Q), 30, 5)
    for h, d in zip(host, dev):
        np.testing.assert_array_equal(np.asarray(h), d.numpy())
    for b in range(len(Q)):
        cs, cc = oracle.search_chunks(E, r2c, Q[b], 30, 5, "dot")
        assert dev[2][b] == len(cc) and dev[1][b, : len(cc)].tolist() == cc.tolist()


def test_bench_refuses_gpus_n_without_n_devices():
    """bench.py --gpus N must become N ranks or fail: here (no GPU at all) `--gpus 2` exits non-zero before any work, and a
    WORLD_SIZE that disagrees with --gpus is refused as well."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        import torch

        if torch.cuda.device_count() >= 2:
            pytest.skip("needs a box with fewer than two GPUs")
    except ImportError:
        pytest.skip("no torch")
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "visible" in res.stderr
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], cwd=root,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
