"""The oracle and the host mirror against the golden vectors produced by the REFERENCE's own
`_embed.py` (oracle/make_golden.py).  CPU only: pooling arithmetic here is the oracle's NumPy;
the GPU run of the same fixtures is tests/test_gpu_parity.py::test_embed_golden_*.
"""

import json

import numpy as np
import pytest

from oracle import oracle
from oracle.fake_embedder import FakeLlama, make_sentences
from raglite_amd import _embed as mirror


def _manifest(golden_dir):
    return json.loads((golden_dir / "manifest.json").read_text())


def _cases(kind):
    from pathlib import Path

    man = json.loads((Path(__file__).parent / "golden" / "manifest.json").read_text())
    return [k for k, v in man.items() if v["kind"] == kind]


def _oracle_late_chunking(sentences, emb, normalize):
    """The oracle's end-to-end restatement of `_embed.py:16-141` (token counting via the mirror's
    sentinel logic is checked separately below; here it is restated in the reference's own shape)."""
    sentinel_tokens = [t for t in emb.tokenize(f"A⊕B ⊕ C.\n⊕D".encode(), add_bos=False)
                       if "⊕" in emb.detokenize([t]).decode()]
    num_tokens_list, batch, batch_len = [], [], 0
    for i, s in enumerate(sentences):
        batch.append(s)
        batch_len += len(s)
        if i == len(sentences) - 1 or batch_len > (emb.n_ctx() // 2):
            toks = np.asarray(emb.tokenize("⊕".join(batch).encode(), add_bos=False), dtype=np.intp)
            for st in sentinel_tokens[1:]:
                toks[toks == st] = sentinel_tokens[0]
            idx = np.where(toks == sentinel_tokens[0])[0]
            num_tokens_list.extend(np.diff(idx, prepend=0, append=len(toks)).tolist())
            batch, batch_len = [], 0
    num_tokens = np.asarray(num_tokens_list, dtype=np.intp)
    rows = []
    for s0, c0, e0 in oracle.create_segments(num_tokens, emb.n_ctx(), emb.n_batch):
        seg = np.asarray(emb.embed("".join(sentences[s0:e0])))
        rows.append(oracle.pool_segment(seg, num_tokens[s0:e0], c0 - s0))
    x = np.vstack(rows)
    if normalize:
        x = oracle.l2_normalize(x)
    return oracle.to_fp16(x), num_tokens


@pytest.mark.parametrize("name", _cases("late_chunking"))
def test_oracle_reproduces_reference_late_chunking(golden_dir, name):
    meta = _manifest(golden_dir)[name]
    golden = np.load(golden_dir / f"{name}.npz")["output"]
    emb = FakeLlama(dim=meta["dim"], n_ctx=meta["n_ctx"], n_batch=meta["n_batch"], seed=meta["embedder_seed"])
    sentences = make_sentences(meta["sentence_seed"], meta["n_sentences"])
    out, _ = _oracle_late_chunking(sentences, emb, meta["normalize"])
    assert out.dtype == np.float16 and out.shape == golden.shape
    assert np.array_equal(out.view(np.uint16), golden.view(np.uint16)), "oracle differs from the reference bit-wise"


@pytest.mark.parametrize("name", _cases("late_chunking"))
def test_host_mirror_spans_reproduce_reference(golden_dir, name):
    """The mirror's own bookkeeping (token counts, segment plan, row split) + oracle pooling on its spans
    must give the reference's bits: pins everything in raglite_amd/_embed.py except the kernel."""
    meta = _manifest(golden_dir)[name]
    golden = np.load(golden_dir / f"{name}.npz")["output"]
    emb = FakeLlama(dim=meta["dim"], n_ctx=meta["n_ctx"], n_batch=meta["n_batch"], seed=meta["embedder_seed"])
    sentences = make_sentences(meta["sentence_seed"], meta["n_sentences"])
    tokens, b, e = mirror.plan_document(sentences, emb)
    assert tokens.dtype == np.float32 and len(b) == len(e) == len(sentences)
    assert emb.embed_calls == meta["embed_calls"]  # same number of segments as the reference made
    _, out = oracle.pool_norm_cast(tokens, b, e, normalize=meta["normalize"])
    assert np.array_equal(out.view(np.uint16), golden.view(np.uint16))


@pytest.mark.parametrize("name", _cases("batch"))
def test_oracle_reproduces_reference_batch_pool(golden_dir, name):
    meta = _manifest(golden_dir)[name]
    golden = np.load(golden_dir / f"{name}.npz")["output"]
    emb = FakeLlama(dim=meta["dim"], n_ctx=meta["n_ctx"], seed=meta["embedder_seed"])
    strings = make_sentences(meta["sentence_seed"], meta["n_sentences"])
    out = oracle.embed_string_batch_pool([emb.token_matrix(s) for s in strings], normalize=meta["normalize"])
    assert np.array_equal(out.view(np.uint16), golden.view(np.uint16))
    # span form with the eps guard (what the kernel computes for this path)
    mats = [emb.token_matrix(s) for s in strings]
    ends = np.cumsum([len(m) for m in mats])
    begins = ends - np.asarray([len(m) for m in mats])
    _, out2 = oracle.pool_norm_cast(np.vstack(mats), begins, ends, normalize=meta["normalize"],
                                    eps=float(np.finfo(np.float64).eps))
    assert np.array_equal(out2.view(np.uint16), golden.view(np.uint16))


def test_reference_behavioural_contract(golden_dir):
    """`tests/test_embed.py:19-26` of the reference: fp16, finite, unit norm (rtol 1e-3)."""
    for name, meta in _manifest(golden_dir).items():
        g = np.load(golden_dir / f"{name}.npz")["output"]
        assert g.dtype == np.float16 and np.all(np.isfinite(g)) and len(g) == meta["n_sentences"]
        if meta["normalize"]:
            assert np.allclose(np.linalg.norm(g.astype(np.float64), axis=1), 1.0, rtol=1e-3)


def test_plan_segments_matches_oracle_random():
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 80))
        toks = rng.integers(0, 60, size=n).astype(np.intp)
        n_ctx = int(rng.integers(80, 600))
        if toks.max() > min(n_ctx, n_ctx) - 16 - round(0.382 * (n_ctx - 16)):
            continue  # the reference loops forever on an over-long sentence; the mirror documents its deviation
        assert mirror.plan_segments(toks, n_ctx, n_ctx) == oracle.create_segments(toks, n_ctx, n_ctx)


def test_split_rows_matches_oracle_random():
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = int(rng.integers(1, 40))
        toks = rng.integers(1, 50, size=n).astype(np.intp)
        rows = int(toks.sum() + rng.integers(0, 5))
        a, b = mirror.split_rows(rows, toks), oracle.largest_remainder_sizes(rows, toks)
        assert np.array_equal(a, b) and a.sum() == rows


def test_plan_segments_oversized_sentence_terminates():
    plan = mirror.plan_segments(np.asarray([5, 900, 7], dtype=np.intp), 128, 128)
    assert [p[1:] for p in plan] == [(0, 1), (1, 2), (2, 3)]


# ---------------------------------------------------------------------------------------------------
# 8f-3: the query adapter's producer, pinned against the reference's own lines
# ---------------------------------------------------------------------------------------------------
def test_query_adapter_arithmetic_matches_reference_golden():
    """tests/golden/query_adapter.npz was produced by executing `_query_adapter.py`'s own source text
    (oracle/make_golden_adapter.py): target optimisation, the closed-form adapter, positive/negative row choice."""
    from pathlib import Path

    g = np.load(Path(__file__).parent / "golden" / "query_adapter.npz")
    for i in range(int(g["n_target_cases"])):
        t = oracle.optimize_query_target(g[f"target{i}_q"], g[f"target{i}_P"], g[f"target{i}_N"], float(g[f"target{i}_alpha"]))
        assert t.dtype == g[f"target{i}_t"].dtype
        assert np.array_equal(t.view(np.uint16), g[f"target{i}_t"].view(np.uint16))  # same solver, same inputs: same bits
    for i in range(int(g["n_adapter_cases"])):
        A = oracle.query_adapter_from_targets(g[f"adapter{i}_Q"], g[f"adapter{i}_T"], str(g[f"adapter{i}_metric"]))
        np.testing.assert_allclose(A, g[f"adapter{i}_A"], rtol=0, atol=1e-12)
        if str(g[f"adapter{i}_metric"]) == "cosine":
            np.testing.assert_allclose(A @ A.T, np.eye(len(A)), atol=1e-10)  # orthogonal Procrustes solution
    E, q = g["select_E"], g["select_q"]
    assert np.array_equal(E[[oracle.best_row(E, q)]], g["select_row"])


def test_reciprocal_rank_fusion_matches_reference_golden():
    """`_search.py:233-252` executed from its own source text (oracle/make_golden_adapter.py) vs the host mirror."""
    import json
    from pathlib import Path

    import raglite_amd

    g = np.load(Path(__file__).parent / "golden" / "query_adapter.npz")
    for case in json.loads(str(g["rrf_json"])):
        ids, scores = raglite_amd.reciprocal_rank_fusion(case["rankings"], k=case["k"], weights=case["weights"])
        assert ids == case["ids"] and scores == case["scores"]  # same float operations in the same order
    with pytest.raises(ValueError):
        raglite_amd.reciprocal_rank_fusion([["a"]], weights=[1.0, 2.0])


# ---------------------------------------------------------------------------------------------------
# 8f-4: semantic chunking, pinned against the reference's own split_chunks
# ---------------------------------------------------------------------------------------------------
def _split_cases():
    import json
    from pathlib import Path

    g = np.load(Path(__file__).parent / "golden" / "split_chunks.npz")
    meta = json.loads(str(g["meta_json"]))
    return [(m["chunklets"], g[f"case{i}_X"], m["max_size"], g[f"case{i}_cost"], g[f"case{i}_sizes"], m["chunks"])
            for i, m in enumerate(meta)]


def test_partition_similarity_oracle_matches_reference_cost_vector():
    """The MILP cost vector the REAL `split_chunks` handed to linprog (captured by oracle/make_golden_chunks.py)."""
    for chunklets, X, _max_size, cost, sizes, _chunks in _split_cases():
        if len(cost) == 0:  # single-chunk early exit: the reference never computes similarities
            assert len(sizes) == 1
            continue
        lens = np.asarray([len(c) for c in chunklets])
        got = oracle.heading_adjusted(oracle.partition_similarity(X, lens), chunklets)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, cost, rtol=0, atol=1e-6)


def test_partition_milp_mirror_reproduces_reference_chunks():
    """Host half of the mirror: given the reference's cost vector, the same partition comes out."""
    from raglite_amd._chunking import _solve_partition

    for chunklets, _X, max_size, cost, sizes, chunks in _split_cases():
        if len(cost) == 0:
            continue
        lens = np.asarray([len(c) for c in chunklets])
        cuts = _solve_partition(cost.astype(np.float32), lens, max_size)
        bounds = [0, *cuts, len(chunklets)]
        assert [j - i for i, j in zip(bounds[:-1], bounds[1:])] == sizes.tolist()
        assert ["".join(chunklets[i:j]) for i, j in zip(bounds[:-1], bounds[1:])] == chunks


def test_adapter_application_and_num_hits_match_reference_lines():
    """a5 (`_search.py:57-62`) and the num_hits rule (`:66-67`), exec'd from vector_search's own body: the oracle and
    the host mirror's arithmetic reproduce them exactly."""
    from pathlib import Path

    from raglite_amd import _search

    g = np.load(Path(__file__).parent / "golden" / "query_adapter.npz")
    for i in range(int(g["n_a5_cases"])):
        A, q, want = g[f"a5_{i}_A"], g[f"a5_{i}_q"], g[f"a5_{i}_out"]
        got = oracle.adapter_apply(A, q)
        assert got.dtype == want.dtype and np.array_equal(got, want)
    for oversample, chunk_max_size, num_results, want in g["num_hits_cases"].tolist():
        assert oracle.num_hits(num_results, oversample, chunk_max_size) == want


def test_chunk_text_matches_the_reference_str_chunk(golden_dir):
    """`str(chunk)` -- what `rerank_chunks` hands the reranker (`_search.py:394-396`) and what `MaxSimRanker.rank(docs=...)` maps
    back to chunk ordinals -- against strings produced by the reference's own `Chunk.front_matter` / `.content`
    (`_database.py:300-320`, oracle/make_golden_chunktext.py): list-valued metadata prints as a list, `url: [None]` is a line."""
    import json

    from raglite_amd._store import chunk_text

    cases = json.loads((golden_dir / "chunk_text.json").read_text())
    assert len(cases) >= 6
    for case in cases:
        assert chunk_text(case["headings"], case["body"], case["metadata"]) == case["text"]
    assert "filename: ['a.md']\nurl: [None]" in cases[0]["text"]
