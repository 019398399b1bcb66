"""f2 with a REAL SentencePiece tokenizer: the reference counts every sentence's tokens with the embedding
model's own tokenizer through a sentinel trick (`/root/reference/src/raglite/_embed.py:20-36,64-93`).  `tests/golden/spm_token_counts.json`
holds what the reference's OWN statements (exec'd from its source by `oracle/make_golden_tokens.py`) count over a SentencePiece BPE
model (`tests/golden/spm_2k.model`; bge-m3's 250 k-piece `sentencepiece.bpe.model` is not fetchable here); here the same model is loaded
through the path a bge-m3 tokenizer file would take -- `raglite_amd.SentencePieceTokenizer` behind `TorchTokenEmbedder` -- and the
mirror's `count_sentence_tokens` must reproduce those counts exactly.  `HashTokenizer` stays the fallback, not the only tokenizer under test."""

import json
from pathlib import Path

import numpy as np
import pytest

spm = pytest.importorskip("sentencepiece")

from oracle.fake_embedder import SENTINEL, make_sentences  # noqa: E402
from raglite_amd import EncoderShape, SentencePieceTokenizer, TorchTokenEmbedder  # noqa: E402
from raglite_amd._embed import _sentinel_token_ids, count_sentence_tokens, plan_segments  # noqa: E402

GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def embedder():
    spec = json.loads((GOLDEN / "spm_token_counts.json").read_text())
    tok = SentencePieceTokenizer(GOLDEN / spec["model"])
    shape = EncoderShape(vocab_size=tok.vocab_size, hidden=32, layers=1, heads=2, ffn=64, max_positions=spec["n_ctx"] + 8, n_ctx=spec["n_ctx"])
    return TorchTokenEmbedder(shape, tokenizer=tok, device="cpu"), spec


def test_xlmr_id_layout(embedder):
    emb, _ = embedder
    tok = emb.tokenizer
    assert tok.vocab_size == 2001
    ids = tok.encode("Hello world. " + SENTINEL + " zzz")
    assert all(3 <= i < tok.vocab_size for i in ids)  # <s> / <pad> / </s> are never produced by text
    assert tok.encode("中") == [4, 3] or 3 in tok.encode("中")  # a character outside the alphabet is <unk> = 3
    assert tok.decode(tok.encode("abc def.")) == "abc def."
    assert emb.tokenize(b"abc", add_bos=True)[0] == 0 and emb.tokenize(b"abc", add_bos=False) == tok.encode("abc")
    raw = spm.SentencePieceProcessor(model_file=str(GOLDEN / "spm_2k.model"))
    assert [i - 1 for i in tok.encode("abc def.")] == raw.encode("abc def.")  # the fairseq offset, nothing else


def test_token_counts_equal_the_reference_code_over_the_same_tokenizer(embedder):
    emb, spec = embedder
    for name, case in spec["cases"].items():
        sentences = make_sentences(case["seed"], case["n"])
        assert _sentinel_token_ids(emb) == case["sentinel_tokens"], name
        counts = count_sentence_tokens(sentences, emb)
        assert counts.tolist() == case["num_tokens"], name
        # the counts feed the segment plan: every segment fits the context the reference allows (`_embed.py:94-110`)
        for seg_start, content_start, seg_end in plan_segments(counts, emb.n_ctx(), emb.n_batch):
            assert 0 <= seg_start <= content_start < seg_end <= len(sentences)
            assert counts[seg_start:seg_end].sum() <= emb.n_ctx() - 16 or seg_end - content_start == 1


def test_forward_over_sentencepiece_ids(embedder):
    emb, _ = embedder
    mats = emb.embed(["Abc def ghi. ", "Xyz. "])
    for text, m in zip(["Abc def ghi. ", "Xyz. "], mats):
        assert m.shape == (len(emb.tokenizer.encode(text)) + 2, 32) and bool(np.isfinite(m.numpy()).all())
