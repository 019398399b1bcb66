"""Generate `tests/golden/spm_2k.model` and `tests/golden/spm_token_counts.json`: the reference's OWN token-count code run over a real
SentencePiece tokenizer -- TEST INFRASTRUCTURE.  Run in the authoring container only (needs /root/reference and `sentencepiece`):

    python -m oracle.make_golden_tokens

The reference counts the tokens of every sentence with the embedding model's tokenizer through a sentinel trick
(`src/raglite/_embed.py:20-36` `_count_tokens`, `:68-77` sentinel detection, `:83-93` batching by n_ctx // 2 characters).  Those
statements sit inside `embed_strings_with_late_chunking`, which also needs llama.cpp; so the REAL source text is cut out with `ast`
and exec'd piecewise (nothing is copied into this repository), against a llama-like facade over a SentencePiece BPE model:

* the model: 2 000 pieces, BPE (bge-m3's `sentencepiece.bpe.model` is a BPE model of 250 k pieces -- not fetchable here), trained on
  `oracle.fake_embedder.make_sentences` text that also contains the sentinel character, so that the vocabulary knows it (as XLM-R's
  does).  Training is done ONCE and the model file is committed: the fixture does not depend on SentencePiece's trainer being
  reproducible;
* ids in XLM-RoBERTa's layout (SentencePiece id + 1, `<unk>` = 3), which is what `raglite_amd.SentencePieceTokenizer` implements --
  the facade here does the same arithmetic independently, on the raw `sentencepiece` processor.
"""

from __future__ import annotations

import ast
import io
import json
import textwrap
from pathlib import Path

import numpy as np

from oracle.fake_embedder import SENTINEL, make_sentences

SRC = Path("/root/reference/src/raglite/_embed.py")
GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"
MODEL = GOLDEN / "spm_2k.model"
OUT = GOLDEN / "spm_token_counts.json"
N_CTX = 512


def _train() -> bytes:
    import sentencepiece as spm

    text = make_sentences(4101, 6000)
    # the sentinel in every context the reference's probe string uses (`_embed.py:71`), so that BPE may or may not merge it with a neighbour
    text += [f"A{SENTINEL}B {SENTINEL} C.\n{SENTINEL}D"] * 50 + [f"{a}{SENTINEL}{b}" for a, b in zip(text[:400], text[400:800])]
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(text), model_writer=model, vocab_size=2000, model_type="bpe", character_coverage=1.0,
                                   num_threads=1, unk_id=0, bos_id=1, eos_id=2, pad_id=-1, input_sentence_size=0, shuffle_input_sentence=False)
    return model.getvalue()


class SpmLlama:
    """What `_embed.py` touches of a `llama_cpp.Llama`, over a raw SentencePiece processor in XLM-R's id layout."""

    def __init__(self, proto: bytes) -> None:
        import sentencepiece as spm

        self.sp = spm.SentencePieceProcessor(model_proto=proto)
        self.n_batch = N_CTX

    def n_ctx(self) -> int:
        return N_CTX

    def tokenize(self, data: bytes, add_bos: bool = True, special: bool = False) -> list[int]:  # noqa: ARG002,FBT001,FBT002
        ids = [i + 1 if i else 3 for i in self.sp.encode(data.decode())]
        return ([0] if add_bos else []) + ids

    def detokenize(self, tokens: list[int]) -> bytes:
        return self.sp.decode([t - 1 for t in tokens if t > 3]).encode()


def _reference_token_counts(embedder, sentences: list[str]):
    """(sentinel_tokens, num_tokens) by the reference's statements, exec'd from its source."""
    text = SRC.read_text()
    tree = ast.parse(text)
    outer = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "embed_strings_with_late_chunking")
    count_fn = next(n for n in outer.body if isinstance(n, ast.FunctionDef) and n.name == "_count_tokens")
    lines = text.splitlines()
    fn_src = textwrap.dedent("\n".join(lines[count_fn.lineno - 1 : count_fn.end_lineno]))
    ns = {"np": np, "Llama": object}
    exec(compile(fn_src, str(SRC), "exec"), ns)  # noqa: S102 - the reference's own code, read-only
    a = next(i for i, ln in enumerate(lines) if "# Identify the tokens corresponding to a sentinel character." in ln)
    b = next(i for i, ln in enumerate(lines) if "# Compute the maximum number of tokens for each segment's preamble and content." in ln)
    block = textwrap.dedent("\n".join(lines[a:b]))
    env = {"np": np, "embedder": embedder, "sentences": sentences, "n_ctx": embedder.n_ctx(), "_count_tokens": ns["_count_tokens"]}
    exec(compile(block, str(SRC), "exec"), env)  # noqa: S102
    return [int(t) for t in env["sentinel_tokens"]], [int(x) for x in env["num_tokens"]]


def main() -> None:
    if not MODEL.exists():
        MODEL.write_bytes(_train())
    emb = SpmLlama(MODEL.read_bytes())
    cases = {}
    for name, seed, n in (("prose_60", 4201, 60), ("prose_400", 4202, 400), ("single", 4203, 1)):
        sentences = make_sentences(seed, n)
        sentinel_tokens, counts = _reference_token_counts(emb, sentences)
        cases[name] = {"seed": seed, "n": n, "sentinel_tokens": sentinel_tokens, "num_tokens": counts}
    OUT.write_text(json.dumps({"n_ctx": N_CTX, "model": MODEL.name, "cases": cases}, indent=1) + "\n")
    totals = ", ".join(f"{k}: {sum(v['num_tokens'])} tokens" for k, v in cases.items())
    print(f"wrote {OUT} ({totals}); model {MODEL.stat().st_size} bytes")


if __name__ == "__main__":
    main()
