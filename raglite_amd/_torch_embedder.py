"""Token-level embedding forward in PyTorch-ROCm.

The reference gets one embedding per TOKEN from llama.cpp -- `Llama(embedding=True, pooling_type=NONE)`,
`src/raglite/_embed.py:64-66,119,151-154` -- for its default embedder bge-m3, an XLM-RoBERTa-large encoder
(24 layers, d = 1024, 16 heads, FFN 4096).  This module is that forward pass on the GPU behind the same duck-typed
surface `_embed.py` touches (`n_ctx()`, `n_batch`, `tokenize`, `detokenize`, `embed`), so that

    raglite_amd.set_embedder_factory(lambda cfg: TorchTokenEmbedder.bge_m3_shaped(device="cuda"))

makes `embed_strings()` run tokenise -> encoder -> late-chunking pool (`rl_pool_norm`) without the token matrix ever
leaving HBM.  PyTorch is plumbing here (GEMMs through hipBLASLt, attention through SDPA); the kernels this package
owns start at the pooling seam.

No checkpoint can be fetched in this environment: weights are random-initialised in the architecture's shape unless a
state dict is supplied (`load_state_dict`).  Tokenizers: `SentencePieceTokenizer(model_file)` loads a SentencePiece model the way
bge-m3's `sentencepiece.bpe.model` is used (XLM-R id layout; tests/test_spm_tokenizer.py runs the reference's own token-count code on
it); without a model file the fallback is `HashTokenizer`, a deterministic hashing word-piece splitter.  Anything with
`encode(str) -> list[int]` / `decode(list[int]) -> str` can be passed as `tokenizer=`.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import Any

SENTINEL = "⊕"  # `_embed.py:70`


@dataclass(frozen=True)
class EncoderShape:
    vocab_size: int = 250_002
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    ffn: int = 4096
    max_positions: int = 8194
    n_ctx: int = 8192
    layer_norm_eps: float = 1e-5
    pad_id: int = 1  # XLM-R: <s>=0, <pad>=1, </s>=2; positions start at pad_id + 1
    bos_id: int = 0
    eos_id: int = 2
    type_vocab: int = 1  # XLM-R has one segment type whose embedding is added to every token; BERT has two
    position_offset: int = 2  # XLM-R / RoBERTa: position of token t is t + pad_id + 1; BERT: t
    classifier: bool = False  # BERT pooler (dense + tanh on the first token) + a one-logit head (`_cross_encoder.py`)


class HashTokenizer:
    """Deterministic stand-in for a SentencePiece vocabulary: pieces of up to 4 alphanumerics, single other
    characters, the sentinel as its own token; ids by FNV-1a into the vocabulary, remembered for `decode`."""

    def __init__(self, vocab_size: int, reserved: int = 8) -> None:
        self.vocab_size, self.reserved = vocab_size, reserved
        self._piece_of: dict[int, str] = {}
        self._id_of: dict[str, int] = {}
        self._ids_of_run: dict[str, tuple[int, ...]] = {}

    @staticmethod
    def _pieces(text: str) -> list[str]:
        out, i = [], 0
        while i < len(text):
            if text[i].isalnum():
                j = i
                while j < len(text) and j - i < 4 and text[j].isalnum():
                    j += 1
                out.append(text[i:j])
                i = j
            else:
                out.append(text[i])
                i += 1
        return out

    def _id(self, p: str) -> int:
        tid = self._id_of.get(p)
        if tid is None:
            h = 2166136261
            for b in p.encode():
                h = ((h ^ b) * 16777619) & 0xFFFFFFFF
            tid = self.reserved + h % (self.vocab_size - self.reserved)
            for _ in range(64):  # keep decode a function under hash collisions (linear probing, bounded: a table that is filling up --
                if self._piece_of.get(tid, p) == p:  # more distinct pieces than ids -- lets pieces share an id instead of probing for ever)
                    break
                tid = self.reserved + (tid + 1 - self.reserved) % (self.vocab_size - self.reserved)
            self._piece_of.setdefault(tid, p)
            self._id_of[p] = tid
        return tid

    def encode(self, text: str) -> list[int]:
        """The ids of `_pieces(text)`, with the pieces of every alphanumeric run remembered per run (natural text repeats its words: a
        SentencePiece model tokenises in C++, this stand-in should not be what an embedding benchmark measures)."""
        ids: list[int] = []
        for alnum, grp in itertools.groupby(text, key=str.isalnum):
            run = "".join(grp)
            if alnum:
                got = self._ids_of_run.get(run)
                if got is None:
                    got = tuple(self._id(run[i : i + 4]) for i in range(0, len(run), 4))
                    if len(self._ids_of_run) < 1_000_000:
                        self._ids_of_run[run] = got
                ids.extend(got)
            else:
                ids.extend(self._id(c) for c in run)
        return ids

    def decode(self, ids: list[int]) -> str:
        return "".join(self._piece_of.get(t, "") for t in ids)


class SentencePieceTokenizer:
    """A SentencePiece model behind the `encode` / `decode` surface `TorchTokenEmbedder` expects, in XLM-RoBERTa's id layout -- the one
    bge-m3's published `sentencepiece.bpe.model` is used with: `<s>` = 0, `<pad>` = 1, `</s>` = 2, `<unk>` = 3, every other piece at its
    SentencePiece id + 1 (the fairseq offset).  This is the loading path a real bge-m3 tokenizer file takes:

        TorchTokenEmbedder.bge_m3_shaped(tokenizer=SentencePieceTokenizer("sentencepiece.bpe.model"))

    The reference counts tokens with the embedding model's own tokenizer (`src/raglite/_embed.py:64-93`, the sentinel trick of
    `:20-36`); with this class the same calls run through `sentencepiece` instead of the hashing stand-in."""

    FAIRSEQ_OFFSET = 1
    UNK_ID = 3

    def __init__(self, model_file: str | None = None, *, model_proto: bytes | None = None) -> None:
        import sentencepiece as spm

        if (model_file is None) == (model_proto is None):
            raise ValueError("SentencePieceTokenizer needs exactly one of model_file / model_proto")
        self.sp = spm.SentencePieceProcessor(model_file=str(model_file)) if model_file is not None else spm.SentencePieceProcessor(model_proto=model_proto)
        if self.sp.unk_id() != 0:
            raise ValueError("expected a SentencePiece model with <unk> = 0 (the XLM-RoBERTa / bge-m3 layout)")

    @property
    def vocab_size(self) -> int:
        return len(self.sp) + self.FAIRSEQ_OFFSET

    def encode(self, text: str) -> list[int]:
        return [i + self.FAIRSEQ_OFFSET if i else self.UNK_ID for i in self.sp.encode(text)]

    def decode(self, ids: list[int]) -> str:
        return self.sp.decode([i - self.FAIRSEQ_OFFSET for i in ids if i > self.UNK_ID])


def _build_encoder(shape: EncoderShape):
    import torch
    from torch import nn
    from torch.nn import functional as F  # noqa: N812

    class Layer(nn.Module):
        def __init__(self) -> None:
            super().__init__()
            d = shape.hidden
            self.qkv = nn.Linear(d, 3 * d)
            self.out = nn.Linear(d, d)
            self.ln1 = nn.LayerNorm(d, eps=shape.layer_norm_eps)
            self.up = nn.Linear(d, shape.ffn)
            self.down = nn.Linear(shape.ffn, d)
            self.ln2 = nn.LayerNorm(d, eps=shape.layer_norm_eps)

        def forward(self, x, mask, kv_len=None):  # x: (B, T, d); mask: (B, 1, 1, T) additive or None
            B, T, d = x.shape  # noqa: N806
            h = shape.heads
            q, k, v = self.qkv(x).view(B, T, 3, h, d // h).permute(2, 0, 3, 1, 4)  # each (B, h, T, d/h)
            if kv_len is not None:  # rows past kv_len are QUERY padding (see Encoder.forward): nobody attends to them
                k, v = k[:, :, :kv_len], v[:, :, :kv_len]
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            x = self.ln1(x + self.out(a.transpose(1, 2).reshape(B, T, d)))  # post-LN, like BERT / XLM-R
            return self.ln2(x + self.down(F.gelu(self.up(x))))

    class Encoder(nn.Module):
        def __init__(self) -> None:
            super().__init__()
            self.tok = nn.Embedding(shape.vocab_size, shape.hidden, padding_idx=shape.pad_id)
            self.pos = nn.Embedding(shape.max_positions, shape.hidden)
            self.typ = nn.Embedding(shape.type_vocab, shape.hidden)
            self.ln = nn.LayerNorm(shape.hidden, eps=shape.layer_norm_eps)
            self.layers = nn.ModuleList(Layer() for _ in range(shape.layers))
            if shape.classifier:
                self.pooler = nn.Linear(shape.hidden, shape.hidden)
                self.head = nn.Linear(shape.hidden, 1)

        def forward(self, ids, lengths, type_ids=None, *, all_full=None):  # ids: (B, T) padded with pad_id; lengths: (B,)
            """all_full: the caller KNOWS (from host-side lengths) whether every row is T tokens long -- no device round trip to find out;
            None: ask the device (one synchronisation).

            One unpadded sequence (B == 1, the late-chunking segment of `_embed.py:119`) on a GPU: the flash kernels torch ships run the
            same attention 1.36 x faster when the QUERY length is a multiple of 128 (measured on MI355X at 16 x 64, bf16: 7 778 queries
            0.572 ms, 7 808 queries over the same 7 778 keys 0.420 ms; a key-padding mask instead costs 3.3 x), so the sequence is padded to
            the next multiple with pad tokens whose rows only ever act as queries -- keys and values stay the T real tokens, no mask --
            and are cut off at the end: the real tokens' outputs depend on real tokens only."""
            B, T = ids.shape  # noqa: N806
            ar = torch.arange(T, device=ids.device)
            valid = ar[None, :] < lengths[:, None]
            if all_full is None:
                all_full = bool(valid.all())
            pad_q = (-T) % 128 if (B == 1 and all_full and ids.is_cuda and not shape.classifier) else 0
            if pad_q:
                ids = F.pad(ids, (0, pad_q), value=shape.pad_id)
                ar = torch.arange(T + pad_q, device=ids.device)
                valid = ar[None, :] < lengths[:, None]
                if type_ids is not None:
                    type_ids = F.pad(type_ids, (0, pad_q), value=0)
            pos = torch.where(valid, ar[None, :] + shape.position_offset, torch.zeros_like(ids))
            typ = self.typ.weight[0] if type_ids is None else self.typ(type_ids)
            x = self.ln(self.tok(ids) + self.pos(pos) + typ)
            mask = None
            if not all_full:
                mask = torch.zeros((B, 1, 1, T), dtype=x.dtype, device=x.device).masked_fill(~valid[:, None, None, :],
                                                                                           float("-inf"))
            for layer in self.layers:
                x = layer(x, mask, T if pad_q else None)
            if pad_q:
                x = x[:, :T]
            if shape.classifier:
                return self.head(torch.tanh(self.pooler(x[:, 0])))[:, 0]  # (B,): one relevance logit per pair
            return x  # (B, T, hidden): one embedding per token, pooling NONE

    return Encoder()


def native_state_from_hf(state: dict, shape: EncoderShape) -> dict:
    """Rename a Hugging Face `XLMRobertaModel` / `BertModel` / `BertForSequenceClassification` state dict (the layout
    bge-m3 and the ms-marco MiniLM cross-encoders are published in) to this module's parameters; q/k/v are fused into
    one (3d, d) projection.  Unknown keys (`position_ids` buffers, an unused pooler) are ignored."""
    import torch

    def strip(k: str) -> str:
        for pre in ("roberta.", "bert.", "model."):
            if k.startswith(pre):
                return k[len(pre):]
        return k

    src = {strip(k): v for k, v in state.items()}
    out = {"tok.weight": src["embeddings.word_embeddings.weight"], "pos.weight": src["embeddings.position_embeddings.weight"],
           "typ.weight": src["embeddings.token_type_embeddings.weight"], "ln.weight": src["embeddings.LayerNorm.weight"],
           "ln.bias": src["embeddings.LayerNorm.bias"]}
    names = {"attention.output.dense": "out", "attention.output.LayerNorm": "ln1", "intermediate.dense": "up",
             "output.dense": "down", "output.LayerNorm": "ln2"}
    for i in range(shape.layers):
        pre = f"encoder.layer.{i}."
        for part in ("weight", "bias"):
            out[f"layers.{i}.qkv.{part}"] = torch.cat([src[f"{pre}attention.self.{n}.{part}"] for n in ("query", "key", "value")])
            for hf, mine in names.items():
                out[f"layers.{i}.{mine}.{part}"] = src[f"{pre}{hf}.{part}"]
    if shape.classifier:
        out.update({"pooler.weight": src["pooler.dense.weight"], "pooler.bias": src["pooler.dense.bias"],
                    "head.weight": src["classifier.weight"], "head.bias": src["classifier.bias"]})
    return out


class TorchTokenEmbedder:
    """llama-like facade over the encoder: the object `embedder_for(config)` is expected to return."""

    def __init__(self, shape: EncoderShape | None = None, *, tokenizer: Any | None = None, device: str = "cuda",
                 dtype: Any | None = None, seed: int = 0, n_batch: int | None = None) -> None:
        import torch

        self.shape = shape or EncoderShape()
        self.tokenizer = tokenizer or HashTokenizer(self.shape.vocab_size)
        self.device = torch.device(device)
        self.dtype = dtype or (torch.bfloat16 if self.device.type == "cuda" else torch.float32)
        self.n_batch = n_batch or self.shape.n_ctx
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)  # random-initialised weights of the architecture's shape (no checkpoint here)
        self.encoder = _build_encoder(self.shape).to(device=self.device, dtype=self.dtype).eval()
        torch.random.set_rng_state(gen_state)
        self.embed_calls = 0

    @classmethod
    def bge_m3_shaped(cls, **kw: Any) -> "TorchTokenEmbedder":
        return cls(EncoderShape(), **kw)

    def load_state_dict(self, state: dict) -> None:
        self.encoder.load_state_dict(state)

    def load_hf_state_dict(self, state: dict) -> None:
        """Weights in the Hugging Face layout bge-m3 is published in (`XLMRobertaModel.state_dict()`)."""
        self.encoder.load_state_dict({k: v.to(self.dtype) for k, v in native_state_from_hf(state, self.shape).items()})

    # -- the surface `_embed.py` touches ---------------------------------------------------------------------
    def n_ctx(self) -> int:
        return self.shape.n_ctx

    def tokenize(self, data: bytes, add_bos: bool = True, special: bool = False) -> list[int]:  # noqa: ARG002,FBT001,FBT002
        ids = self.tokenizer.encode(data.decode())
        return ([self.shape.bos_id] if add_bos else []) + ids

    def detokenize(self, tokens: list[int]) -> bytes:
        return self.tokenizer.decode([t for t in tokens if t not in (self.shape.bos_id, self.shape.eos_id)]).encode()

    def embed(self, text):  # noqa: ANN001,ANN201 - mirrors llama_cpp.Llama.embed(str | list[str])
        """One (T_i, hidden) float32 CUDA tensor per input string (T_i = tokens + <s> + </s>, truncated to n_ctx);
        a single tensor for a single string.  Strings of one call are padded into one batch."""
        import torch

        self.embed_calls += 1
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        rows = []
        for t in texts:
            ids = self.tokenizer.encode(t)[: self.shape.n_ctx - 2]
            rows.append([self.shape.bos_id, *ids, self.shape.eos_id])
        # Lengths stay on the HOST and the ids go up from pinned memory without blocking: nothing in this call waits for the device, so the
        # caller tokenises the next segment while the GPU is still in this one (a `.to(device)` from pageable memory and `int(tensor)`
        # each waited for the whole forward pass: 8 ms of idle GPU per 7.8 k-token segment)
        lens = [len(r) for r in rows]
        T = max(lens)  # noqa: N806
        ids = torch.full((len(rows), T), self.shape.pad_id, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
        lengths = torch.tensor(lens, dtype=torch.long)
        if self.device.type == "cuda":
            ids, lengths = ids.pin_memory().to(self.device, non_blocking=True), lengths.pin_memory().to(self.device, non_blocking=True)
        with torch.inference_mode():
            out = self.encoder(ids, lengths, all_full=min(lens) == T).float()
        mats = [out[i, : lens[i]] for i in range(len(rows))]
        return mats[0] if single else mats
