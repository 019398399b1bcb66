"""Cross-encoder reranker in PyTorch-ROCm ("GPU cross-encoder rerank as an alternative
`BaseRanker`").

The reference's default rerankers are FlashRank cross-encoders -- `FlashRankRanker("ms-marco-MiniLM-L-12-v2")` for
English, `ms-marco-MultiBERT-L-12` otherwise (`src/raglite/_config.py:72-78`) -- called through
`reranker.rank(query=, docs=)` at `src/raglite/_search.py:394-396`.  FlashRank (a third-party dependency, absent from
the reference tree and from this image) runs the ONNX export of a BERT sequence classifier on the CPU, one
`[CLS] query [SEP] passage [SEP]` pair per candidate, and maps the single logit through a sigmoid.  This module is that
forward pass on the GPU behind the same plugin surface: a BERT encoder in the ms-marco-MiniLM-L-12 shape (12 layers,
d = 384, 12 heads, FFN 1536, vocabulary 30 522, 512 positions, two segment types), the pooler (dense + tanh on the
first token) and a one-logit head; all candidates of a query go through the encoder in length-sorted padded batches.

    config.reranker = TorchCrossEncoderRanker.minilm_l12_shaped(device="cuda")     # or {"en": ..., "other": ...}

PyTorch is plumbing here (GEMMs through hipBLASLt, attention through SDPA), as for `_torch_embedder.py`; the module
shares that encoder.  Its arithmetic is pinned against Hugging Face's `BertForSequenceClassification` with the same
weights (`tests/test_host_logic.py`, `tests/test_gpu_parity.py`).  No checkpoint can be fetched in this environment:
weights are random-initialised in the architecture's shape unless `load_hf_state_dict` is given the published ones,
and the default tokenizer is the hashing stand-in; pass `tokenizer=` (anything with `encode(str) -> list[int]`, e.g. the
model's WordPiece `tokenizers.Tokenizer` wrapped to return ids without special tokens) for real text.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Sequence

from raglite_amd._torch_embedder import EncoderShape, HashTokenizer, _build_encoder, native_state_from_hf


@dataclass(frozen=True)
class CrossEncoderShape(EncoderShape):
    """ms-marco-MiniLM-L-12-v2 (BERT, uncased WordPiece vocabulary)."""

    vocab_size: int = 30_522
    hidden: int = 384
    layers: int = 12
    heads: int = 12
    ffn: int = 1536
    max_positions: int = 512
    n_ctx: int = 512
    layer_norm_eps: float = 1e-12
    pad_id: int = 0
    bos_id: int = 101  # [CLS]
    eos_id: int = 102  # [SEP]
    type_vocab: int = 2
    position_offset: int = 0
    classifier: bool = True


def truncate_pair(n_query: int, n_doc: int, budget: int) -> tuple[int, int]:
    """Token counts kept of (query, passage) when the pair exceeds `budget`: the `longest_first` strategy of the
    `tokenizers` library FlashRank truncates with -- drop one token at a time from the longer sequence, from the
    passage on a tie -- in closed form."""
    over = n_query + n_doc - budget
    if over <= 0:
        return n_query, n_doc
    if n_query > n_doc:
        cut = min(over, n_query - n_doc)
        n_query, over = n_query - cut, over - cut
    elif n_doc > n_query:
        cut = min(over, n_doc - n_query)
        n_doc, over = n_doc - cut, over - cut
    n_doc -= (over + 1) // 2
    n_query -= over // 2
    return max(n_query, 0), max(n_doc, 0)


class TorchCrossEncoderRanker:
    """`rank(query=, docs=)` of the `rerankers.BaseRanker` surface the reference calls, on the GPU."""

    def __init__(self, shape: CrossEncoderShape | None = None, *, tokenizer: Any | None = None, device: str = "cuda",
                 dtype: Any | None = None, seed: int = 0, max_length: int | None = None, pairs_per_batch: int = 128) -> None:
        import torch

        self.shape = shape or CrossEncoderShape()
        if not self.shape.classifier:
            raise ValueError("a cross-encoder needs shape.classifier = True")
        self.tokenizer = tokenizer or HashTokenizer(self.shape.vocab_size, reserved=max(self.shape.eos_id, self.shape.bos_id) + 1)
        self.device = torch.device(device)
        self.dtype = dtype or (torch.bfloat16 if self.device.type == "cuda" else torch.float32)
        self.max_length = min(max_length or self.shape.n_ctx, self.shape.n_ctx, self.shape.max_positions - self.shape.position_offset)
        self.pairs_per_batch = pairs_per_batch
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.encoder = _build_encoder(self.shape).to(device=self.device, dtype=self.dtype).eval()
        torch.random.set_rng_state(gen_state)

    @classmethod
    def minilm_l12_shaped(cls, **kw: Any) -> "TorchCrossEncoderRanker":
        return cls(CrossEncoderShape(), **kw)

    def load_hf_state_dict(self, state: dict) -> None:
        """Weights in the layout the ms-marco cross-encoders are published in (`BertForSequenceClassification`)."""
        self.encoder.load_state_dict({k: v.to(self.dtype) for k, v in native_state_from_hf(state, self.shape).items()})

    # ------------------------------------------------------------------------------------------------------
    def encode_pairs(self, query: str, docs: Sequence[str]) -> list[tuple[list[int], int]]:
        """Per doc: the ids of `[CLS] q [SEP] d [SEP]` and the length of the first segment (type 0) inside them."""
        q_ids = list(self.tokenizer.encode(query))
        out = []
        for d in docs:
            d_ids = list(self.tokenizer.encode(d))
            nq, nd = truncate_pair(len(q_ids), len(d_ids), self.max_length - 3)
            out.append(([self.shape.bos_id, *q_ids[:nq], self.shape.eos_id, *d_ids[:nd], self.shape.eos_id], nq + 2))
        return out

    def logits(self, query: str, docs: Sequence[str]):  # noqa: ANN201
        """(len(docs),) float32 tensor of relevance logits on `self.device`."""
        import torch

        pairs = self.encode_pairs(query, docs)
        out = torch.empty(len(pairs), dtype=torch.float32, device=self.device)
        order = sorted(range(len(pairs)), key=lambda i: len(pairs[i][0]))  # length-sorted batches: little padding
        for lo in range(0, len(order), self.pairs_per_batch):
            sel = order[lo : lo + self.pairs_per_batch]
            lengths = torch.tensor([len(pairs[i][0]) for i in sel])
            T = int(lengths.max())  # noqa: N806
            ids = torch.full((len(sel), T), self.shape.pad_id, dtype=torch.long)
            for r, i in enumerate(sel):
                ids[r, : len(pairs[i][0])] = torch.tensor(pairs[i][0], dtype=torch.long)
            first = torch.tensor([pairs[i][1] for i in sel])
            ar = torch.arange(T)
            type_ids = ((ar[None, :] >= first[:, None]) & (ar[None, :] < lengths[:, None])).long()
            with torch.inference_mode():
                z = self.encoder(ids.to(self.device), lengths.to(self.device), type_ids.to(self.device))
            out[torch.tensor(sel, device=self.device)] = z.float()
        return out

    def score(self, query: str, docs: Sequence[str]):  # noqa: ANN201
        """FlashRank's score: sigmoid of the logit, as a NumPy float32 vector in input order."""
        import torch

        return torch.sigmoid(self.logits(query, docs)).cpu().numpy()

    def rank(self, query: str, docs: Sequence[Any], doc_ids: Sequence[int] | None = None, **_: Any):  # noqa: ANN201
        import numpy as np

        from raglite_amd._search import RankedResults, Result

        docs = [d if isinstance(d, str) else str(d) for d in docs]
        if not docs:
            return RankedResults([], query)
        scores = self.score(query, docs)
        key = np.where(np.isnan(scores), -np.inf, scores)
        order = np.lexsort((np.arange(len(docs)), -key))  # best first; equal scores keep input order
        ids = list(doc_ids) if doc_ids is not None else list(range(len(docs)))
        return RankedResults(
            [Result(doc_id=ids[i], score=float(scores[i]), rank=r + 1, text=docs[i]) for r, i in enumerate(order)], query)
