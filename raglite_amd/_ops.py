"""Thin array-level wrappers over the C ABI.

Every function accepts either NumPy arrays (host pointers: the library stages them through HBM
and the call is synchronous) or `torch` CUDA tensors (device pointers: work is enqueued on torch's
current stream and results come back as CUDA tensors, no synchronisation).  PyTorch is only the
owner of device memory and streams here; no torch kernel runs on the hot path.
"""

from __future__ import annotations

import ctypes as C
import threading
from typing import Any

import numpy as np

from raglite_amd import _abi
from raglite_amd._abi import MEM_DEVICE, MEM_HOST, METRICS, SYNTH_KINDS, check, lib

K_MAX = 2048


def _is_torch(x: Any) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _torch():
    import torch

    return torch


def _is_half(x: Any) -> bool:
    """IEEE fp16 data (NumPy float16 or torch.float16)?"""
    if _is_torch(x):
        return x.dtype == _torch().float16
    return getattr(x, "dtype", None) == np.float16


class _Args:
    """Collects the pointers of one call and enforces that they all live on the same side."""

    def __init__(self) -> None:
        self.mem: int | None = None
        self.device = None
        self.keep: list[Any] = []

    def _side(self, mem: int, device=None) -> None:
        if self.mem is None:
            self.mem, self.device = mem, device
        elif self.mem != mem:
            raise ValueError("all array arguments of one call must be NumPy arrays or all CUDA tensors")

    def inp(self, x: Any, dtype: np.dtype) -> int:
        if _is_torch(x):
            torch = _torch()
            if not x.is_cuda:
                raise ValueError("torch tensors passed to raglite_amd must live on a CUDA (HIP) device")
            t = x.to(getattr(torch, np.dtype(dtype).name)).contiguous()
            self._side(MEM_DEVICE, t.device)
            self.keep.append(t)
            return t.data_ptr()
        a = np.ascontiguousarray(x, dtype=dtype)
        self._side(MEM_HOST)
        self.keep.append(a)
        return a.ctypes.data

    def out(self, shape: tuple[int, ...], dtype: np.dtype) -> tuple[Any, int]:
        if self.mem == MEM_DEVICE:
            torch = _torch()
            t = torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), device=self.device)
            self.keep.append(t)
            return t, t.data_ptr()
        a = np.empty(shape, dtype=dtype)
        self.keep.append(a)
        return a, a.ctypes.data

    @property
    def stream(self) -> int:
        if self.mem == MEM_DEVICE:
            return _torch().cuda.current_stream(self.device).cuda_stream
        return 0

    def ensure_device(self) -> None:
        """Make the HIP device current for this thread (ctypes calls run on the caller's thread)."""
        dev = self.device.index if (self.mem == MEM_DEVICE and self.device.index is not None) else _current_device()
        _ensure_init(dev)


_tls = threading.local()


def _current_device() -> int:
    return getattr(_tls, "device", 0)


def _ensure_init(device: int) -> None:
    if getattr(_tls, "initialised", None) != device:
        _abi.init(device)
        _tls.initialised = device
        _tls.device = device


def set_device(device: int) -> None:
    """Select the GPU used by host-array calls made from this thread."""
    _tls.device = device
    _ensure_init(device)


# ----------------------------------------------------------------------------------------------
def set_default_option(name: str, value: int) -> None:
    """`rl_set_default_option`: the start value of a route option for every index created AFTERWARDS in this process."""
    check(lib().rl_set_default_option(_abi.OPTIONS[name], int(value)))


def get_default_option(name: str) -> int:
    v = C.c_int64(0)
    check(lib().rl_get_default_option(_abi.OPTIONS[name], C.byref(v)))
    return int(v.value)


def synth_fill(out, seed: int, start: int = 0, kind: str = "uniform"):
    """Fill a float32 CUDA tensor with the counter-based synthetic stream (oracle-identical bits)."""
    a = _Args()
    ptr = a.inp(out, np.float32)
    if a.mem != MEM_DEVICE or a.keep[0].data_ptr() != out.data_ptr():
        raise ValueError("synth_fill needs a contiguous float32 CUDA tensor")
    a.ensure_device()
    check(lib().rl_synth_fill(ptr, start, out.numel(), seed, SYNTH_KINDS[kind], a.stream))
    return out


def pool_norm(tokens, span_begin, span_end, *, normalize: bool = True, eps: float = 0.0,
              want_f32: bool = False, want_f16: bool = True):
    """a1+a2(+a3): mean-pool token rows [b, e) per span, L2-normalise, cast.  Returns (f32|None, f16|None).

    Mirrors `src/raglite/_embed.py:131-140` (eps == 0) and `:154-164` (eps > 0)."""
    a = _Args()
    p_tok = a.inp(tokens, np.float32)
    p_b = a.inp(span_begin, np.int64)
    p_e = a.inp(span_end, np.int64)
    tok = a.keep[0]
    if tok.ndim != 2:
        raise ValueError("tokens must be a (T, dim) matrix")
    n_rows, dim = int(tok.shape[0]), int(tok.shape[1])
    n_spans = int(a.keep[1].shape[0])
    if int(a.keep[2].shape[0]) != n_spans:
        raise ValueError("span_begin and span_end must have the same length")
    o32, p32 = a.out((n_spans, dim), np.float32) if want_f32 else (None, None)
    o16, p16 = a.out((n_spans, dim), np.float16) if want_f16 else (None, None)
    a.ensure_device()
    check(lib().rl_pool_norm(p_tok, n_rows, dim, p_b, p_e, n_spans, int(normalize), float(eps), p32, p16,
                             a.mem, a.stream))
    return o32, o16


def adapter_apply(A, queries, *, want_f16: bool = False):
    """a5: out[b] = A @ q[b] (`src/raglite/_search.py:62`).  Accepts a vector or a (B, dim) batch."""
    a = _Args()
    p_a = a.inp(A, np.float32)
    q = queries
    single = q.ndim == 1
    if single:
        q = q.reshape(1, -1)
    p_q = a.inp(q, np.float32)
    dim = int(a.keep[0].shape[0])
    if tuple(a.keep[0].shape) != (dim, dim) or int(a.keep[1].shape[1]) != dim:
        raise ValueError("adapter must be (dim, dim) and queries (B, dim)")
    B = int(a.keep[1].shape[0])
    out, p_o = a.out((B, dim), np.float16 if want_f16 else np.float32)
    a.ensure_device()
    check(lib().rl_adapter_apply(p_a, p_q, B, dim, None if want_f16 else p_o, p_o if want_f16 else None,
                                 a.mem, a.stream))
    return out[0] if single else out


def topk(scores, k: int):
    """Exact top-k per row of a (B, n) score matrix by (score desc, index asc)."""
    a = _Args()
    s2 = scores if scores.ndim == 2 else scores.reshape(1, -1)
    p_s = a.inp(s2, np.float32)
    B, n = int(s2.shape[0]), int(s2.shape[1])
    o_s, p_os = a.out((B, k), np.float32)
    o_i, p_oi = a.out((B, k), np.int32)
    a.ensure_device()
    check(lib().rl_topk(p_s, B, n, n, k, p_os, p_oi, a.mem, a.stream))
    return (o_s, o_i) if scores.ndim == 2 else (o_s[0], o_i[0])


def merge_topk(scores, ids, k: int):
    """Merge per-shard lists: scores/ids are (n_lists, B, k_in) -> (B, k)."""
    a = _Args()
    p_s = a.inp(scores, np.float32)
    p_i = a.inp(ids, np.int32)
    n_lists, B, k_in = (int(v) for v in a.keep[0].shape)
    o_s, p_os = a.out((B, k), np.float32)
    o_i, p_oi = a.out((B, k), np.int32)
    a.ensure_device()
    check(lib().rl_merge_topk(p_s, p_i, n_lists, B, k_in, k, p_os, p_oi, a.mem, a.stream))
    return o_s, o_i


def pack_bits(mask) -> np.ndarray:
    """Boolean mask (NumPy / torch, any device) -> little-endian uint32 bitset (bit i of word i // 32)."""
    if _is_torch(mask):
        mask = mask.detach().cpu().numpy()
    m = np.asarray(mask)
    if m.dtype == np.uint32 and m.ndim == 1:
        return np.ascontiguousarray(m)
    b = np.packbits(m.astype(bool).ravel(), bitorder="little")
    pad = (-b.size) % 4
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    return np.ascontiguousarray(b).view(np.uint32)


class DeviceIndex:
    """Device-resident chunk-embedding matrix + chunk CSR: the GPU image of the reference's
    `chunk_embedding` table (`src/raglite/_database.py:403-430`).

    `embeddings`: (n_rows, dim) float32 NumPy array (copied to HBM) or CUDA tensor (borrowed, kept
    alive by this object).  `chunk_offsets`: ascending int64 CSR of length n_chunks+1 (rows of a chunk
    contiguous, `src/raglite/_split_chunks.py:116-122`); None = one row per chunk."""

    def __init__(self, embeddings, chunk_offsets=None, *, metric: str = "cosine", storage: str = "f32") -> None:
        """storage="f16": keep the corpus as IEEE fp16 in HBM (lossless for the reference's data,
        `src/raglite/_embed.py:140`).  float16 input is taken as is, float32 input is rounded to nearest even."""
        if metric not in METRICS:
            raise ValueError(f"Unsupported metric: {metric}")  # wording of src/raglite/_query_adapter.py:207
        if storage not in ("f32", "f16"):
            raise ValueError("storage must be 'f32' or 'f16'")
        self.storage = storage
        a = _Args()
        if storage == "f16":
            if _is_torch(embeddings):
                t = embeddings.to(_torch().float16).contiguous()
                if not t.is_cuda:
                    raise ValueError("torch tensors passed to raglite_amd must live on a CUDA (HIP) device")
                a._side(MEM_DEVICE, t.device)  # noqa: SLF001
                a.keep.append(t)
                p_e = t.data_ptr()
            else:
                h = np.ascontiguousarray(np.asarray(embeddings).astype(np.float16, copy=False))
                a._side(MEM_HOST)  # noqa: SLF001
                a.keep.append(h)
                p_e = h.ctypes.data
        else:
            p_e = a.inp(embeddings, np.float32)
        emb = a.keep[0]
        if emb.ndim != 2:
            raise ValueError("embeddings must be a (n_rows, dim) matrix")
        self.n_rows, self.dim = int(emb.shape[0]), int(emb.shape[1])
        self.metric = metric
        self.mem = a.mem
        self.device = a.device
        self._keep = emb if a.mem == MEM_DEVICE else None
        if chunk_offsets is None:
            off_ptr, n_chunks = None, self.n_rows
            self.chunk_offsets = None
        else:
            off = np.ascontiguousarray(np.asarray(chunk_offsets.cpu() if _is_torch(chunk_offsets) else chunk_offsets),
                                       dtype=np.int64)
            if off.ndim != 1 or off.size < 1:
                raise ValueError("chunk_offsets must be a 1-D array of length n_chunks + 1")
            off_ptr, n_chunks = off.ctypes.data, int(off.size - 1)
            self.chunk_offsets = off
        self.n_chunks = n_chunks
        a.ensure_device()
        handle = C.c_void_p()
        create = lib().rl_index_create_f16 if storage == "f16" else lib().rl_index_create
        check(create(C.byref(handle), p_e, self.n_rows, self.dim, off_ptr, n_chunks, METRICS[metric], a.mem, a.stream))
        self._handle = handle

    def close(self) -> None:
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            lib().rl_index_destroy(h)
        self._keep = None

    def __del__(self) -> None:  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001,S110 - interpreter shutdown
            pass

    # -- helpers -------------------------------------------------------------------------------
    def _queries(self, a: _Args, q) -> tuple[int, int, bool]:
        single = q.ndim == 1
        q2 = q.reshape(1, -1) if single else q
        ptr = a.inp(q2, np.float32)
        if int(a.keep[-1].shape[1]) != self.dim:
            raise ValueError(f"query dimension {int(a.keep[-1].shape[1])} != index dimension {self.dim}")
        return ptr, int(a.keep[-1].shape[0]), single

    def _prep(self, a: _Args) -> None:
        if a.mem == MEM_HOST and self.mem == MEM_DEVICE:
            _ensure_init(self.device.index or 0)
        else:
            a.ensure_device()

    def _filter(self, a: _Args, chunk_filter):
        """chunk_filter (bool mask over chunks, or a packed uint32 bitset) -> pointer on the call's side, or None."""
        if chunk_filter is None:
            return None
        bits = pack_bits(chunk_filter)
        if bits.size != (self.n_chunks + 31) // 32:
            raise ValueError("chunk_filter must have one entry per chunk")
        if a.mem == MEM_DEVICE:
            t = _torch().from_numpy(bits.view(np.int32)).to(a.device)
            a.keep.append(t)
            return t.data_ptr()
        a.keep.append(bits)
        return bits.ctypes.data

    # -- lifecycle ---------------------------------------------------------------
    def append(self, rows, chunk_sizes=None) -> None:
        """Append embedding rows as new chunks (`insert_documents`, `src/raglite/_insert.py:247-272`): existing
        row / chunk ordinals never change.  chunk_sizes: rows per new chunk (None = one chunk per row)."""
        a = _Args()
        p_r = a.inp(rows, np.float32)
        r = a.keep[0]
        if r.ndim != 2 or int(r.shape[1]) != self.dim:
            raise ValueError("rows must be (n_new_rows, dim)")
        n_new = int(r.shape[0])
        if chunk_sizes is None:
            sizes, p_sz, n_new_chunks = None, None, n_new
        else:
            sizes = np.ascontiguousarray(np.asarray(chunk_sizes), dtype=np.int64)
            p_sz, n_new_chunks = sizes.ctypes.data, int(sizes.size)
        self._prep(a)
        check(lib().rl_index_append(self._handle, p_r, n_new, p_sz, n_new_chunks, a.mem, a.stream))
        if self.chunk_offsets is not None or sizes is not None:
            old = self.chunk_offsets if self.chunk_offsets is not None else np.arange(self.n_rows + 1, dtype=np.int64)
            add = sizes if sizes is not None else np.ones(n_new, np.int64)
            self.chunk_offsets = np.concatenate([old, old[-1] + np.cumsum(add)]).astype(np.int64)
        self.n_rows += n_new
        self.n_chunks += n_new_chunks
        self._keep = None  # the index owns its storage after the first append

    def delete_chunks(self, chunk_ordinals) -> None:
        """Tombstone chunks (`delete_documents`, `src/raglite/_delete.py:148-176`): they never match again."""
        c = np.ascontiguousarray(np.asarray(chunk_ordinals), dtype=np.int64).ravel()
        _ensure_init(self.device.index or 0) if self.mem == MEM_DEVICE else _ensure_init(_current_device())
        check(lib().rl_index_delete_chunks(self._handle, c.ctypes.data, int(c.size), None))

    def compact(self) -> np.ndarray:
        """Reclaim tombstoned chunks (`rl_index_compact`): survivors keep their order and are renumbered 0..live-1.
        Returns remap (old n_chunks,) int64: new ordinal of every old chunk, -1 for a deleted one."""
        _ensure_init(self.device.index or 0) if self.mem == MEM_DEVICE else _ensure_init(_current_device())
        remap = np.empty(self.n_chunks, dtype=np.int64)
        n_rows, n_chunks = C.c_int64(0), C.c_int64(0)
        check(lib().rl_index_compact(self._handle, remap.ctypes.data, C.byref(n_rows), C.byref(n_chunks), None))
        if int(n_chunks.value) != self.n_chunks:
            if self.chunk_offsets is not None:
                sizes = np.diff(self.chunk_offsets)[remap >= 0]
                self.chunk_offsets = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
            self.n_rows, self.n_chunks = int(n_rows.value), int(n_chunks.value)
            self._keep = None  # the index owns its (rewritten) storage
        return remap

    def set_option(self, name: str, value: int) -> None:
        """`rl_index_set_option`: choose a route of this index (`_abi.OPTIONS` names the keys; include/raglite_hip.h "options" says what
        each one does).  Results never depend on an option; speed and memory do."""
        check(lib().rl_index_set_option(self._handle, _abi.OPTIONS[name], int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int64(0)
        check(lib().rl_index_get_option(self._handle, _abi.OPTIONS[name], C.byref(v)))
        return int(v.value)

    def options(self, **kv: int):
        """Context manager: set the options, run the block, restore the previous values (what the A/B tests use)."""
        index = self

        class _Ctx:
            def __enter__(self):
                self.old = {k: index.get_option(k) for k in kv}
                for k, v in kv.items():
                    index.set_option(k, v)
                return index

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    index.set_option(k, v)

        return _Ctx()

    def set_exact_fp32(self, exact: bool = True) -> None:
        """Make the MFMA streaming kernel use exact fp32 MFMAs (an ordered fmaf chain) instead of the default fp16
        (hi, lo) split of its fp32 operands (`rl_index_set_arithmetic`, include/raglite_hip.h)."""
        check(lib().rl_index_set_arithmetic(self._handle, 1 if exact else 0))

    @property
    def arithmetic(self) -> str:
        """What the streaming kernel multiplies with: 'fp32_exact', 'f16_split' or 'f16_stored'."""
        m = C.c_int(0)
        check(lib().rl_index_arithmetic(self._handle, C.byref(m)))
        return {1: "fp32_exact", 2: "f16_split", 3: "f16_stored"}[int(m.value)]

    def live(self) -> tuple[int, int]:
        """(live rows, live chunks)."""
        r, c = C.c_int64(0), C.c_int64(0)
        check(lib().rl_index_live(self._handle, C.byref(r), C.byref(c), None))
        return int(r.value), int(c.value)

    def filter_stats(self) -> dict:
        """What the last bound-filtered search on this index did (`rl_index_filter_stats`): kind, queries, candidates per query
        (mean / max), list capacity, and whether the guarded full-precision fallback ran.  Synchronises."""
        out = (C.c_int64 * 6)()
        st = _Args()
        st._side(self.mem, self.device)  # noqa: SLF001
        self._prep(st)
        check(lib().rl_index_filter_stats(self._handle, out, st.stream))
        kind = {0: "none", 1: "maxsim_batch_hi", 2: "rows_hi", 3: "rows_fused", 4: "rows_fused_hi", 5: "maxsim_batch_f16_exact"}[int(out[0])]
        n = int(out[1])
        return {"kind": kind, "queries": n, "candidates_per_query_mean": (int(out[2]) / n if n else 0.0),
                "candidates_per_query_max": int(out[3]), "list_capacity": int(out[4]), "fallback": bool(out[5])}

    def memory(self) -> dict:
        """Device memory of this index in bytes (`rl_index_memory`): the stored rows, the three optional images (0 = not built: the
        device was too full, the index too small for them to pay, or a switch), the scratch grown so far, free / total device memory
        and the headroom an image must leave free to be built."""
        out = (C.c_int64 * 8)()
        check(lib().rl_index_memory(self._handle, out))
        keys = ("rows", "presplit_image", "hi_image", "hi_plane", "scratch", "device_free", "device_total", "image_headroom")
        return {k: int(v) for k, v in zip(keys, out)}

    IMAGES = {"presplit": 1, "hi_image": 2, "hi_plane": 4}  # include/raglite_hip.h: RL_IMAGE_*

    def prepare(self, *images: str) -> tuple[str, ...]:
        """`rl_index_prepare`: build the named lazy images ("presplit", "hi_image", "hi_plane"; none named = all three) NOW, outside the
        hot path, instead of inside the first search whose route reads them (device allocation + one pass over the rows + one stream
        synchronisation in that call).  Returns the images the index holds afterwards -- one that the options, the shape or the free
        memory do not allow is simply absent (the routes over the stored rows answer, same results)."""
        bits = 0
        for name in images or tuple(self.IMAGES):
            if name not in self.IMAGES:
                raise ValueError(f"unknown image {name!r}: one of {sorted(self.IMAGES)}")
            bits |= self.IMAGES[name]
        built = C.c_uint32(0)
        if self.mem == MEM_DEVICE:
            _ensure_init(self.device.index or 0)
            stream = _torch().cuda.current_stream(self.device).cuda_stream
        else:
            _ensure_init(_current_device())
            stream = 0
        check(lib().rl_index_prepare(self._handle, bits, C.byref(built), stream))
        return tuple(name for name, bit in self.IMAGES.items() if built.value & bit)

    # -- a6 + a7 -------------------------------------------------------------------------------
    def search_rows(self, queries, k: int, chunk_filter=None, rank_limit: int | None = None):
        """Exact top-k rows: (scores (B,k) desc, rows (B,k) int32); padding = (-inf, -1).
        chunk_filter: optional bool mask over chunks (the reference's filter-first branch, `_search.py:105-119`).
        rank_limit: the order-first-then-filter branch (`_search.py:120-141`): only the `rank_limit` nearest live rows of
        each query are eligible, the filter applies to those (None / 0: no cut)."""
        a = _Args()
        p_q, B, single = self._queries(a, queries)
        o_s, p_s = a.out((B, k), np.float32)
        o_r, p_r = a.out((B, k), np.int32)
        p_f = self._filter(a, chunk_filter)
        self._prep(a)
        check(lib().rl_search_rows_ranked(self._handle, p_q, B, k, p_f, int(rank_limit or 0), p_s, p_r, a.mem, a.stream))
        return (o_s[0], o_r[0]) if single else (o_s, o_r)

    # -- the order-first cut of a SHARDED corpus, in stages (rl_rank_cut_*; driven by ShardedIndex.search_rows) -------------------
    def rank_cut_begin(self, queries) -> int:
        """Similarities of every live row for `queries` (B, dim), kept by the index until `rank_cut_finish`.  Returns B."""
        a = _Args()
        p_q, B, _ = self._queries(a, queries)
        self._prep(a)
        check(lib().rl_rank_cut_begin(self._handle, p_q, B, a.mem, a.stream))
        self._rank_side = (a.mem, a.device, B)
        return B

    def _rank_args(self) -> tuple[_Args, int]:
        mem, device, B = self._rank_side
        a = _Args()
        a._side(mem, device)  # noqa: SLF001
        self._prep(a)
        return a, B

    def rank_cut_level(self, level: int, rank_limit: int):
        """This shard's histogram of radix level 0..2 (B, 2048) int32 under the prefix the summed previous levels define."""
        a, B = self._rank_args()
        o, p = a.out((B, 2048), np.int32)
        check(lib().rl_rank_cut_level(self._handle, int(level), int(rank_limit), p, a.mem, a.stream))
        return o

    def rank_cut_level_done(self, level: int, hist_sum) -> None:
        """The level's histogram summed over all shards."""
        a, B = self._rank_args()
        p = a.inp(hist_sum, np.int32)
        if tuple(a.keep[-1].shape) != (B, 2048):
            raise ValueError("hist_sum must be (n_queries, 2048)")
        check(lib().rl_rank_cut_level_done(self._handle, int(level), p, a.mem, a.stream))

    def rank_cut_ties(self, rank_limit: int):
        """(B,) int32: this shard's rows whose key is the global threshold key."""
        a, B = self._rank_args()
        o, p = a.out((B,), np.int32)
        check(lib().rl_rank_cut_ties(self._handle, int(rank_limit), p, a.mem, a.stream))
        return o

    def rank_cut_finish(self, rank_limit: int, ties_before, k: int, chunk_filter=None):
        """This shard's top-k inside the global cut: (scores (B, k), rows (B, k) int32, padding (-inf, -1))."""
        a, B = self._rank_args()
        p_t = a.inp(ties_before, np.int32)
        o_s, p_s = a.out((B, k), np.float32)
        o_r, p_r = a.out((B, k), np.int32)
        p_f = self._filter(a, chunk_filter)
        check(lib().rl_rank_cut_finish(self._handle, int(rank_limit), p_t, p_f, int(k), p_s, p_r, a.mem, a.stream))
        return o_s, o_r

    # -- a6 + a7 + a8 ----------------------------------------------------------------------------
    def search_chunks(self, queries, num_hits: int, k: int, chunk_filter=None, rank_limit: int | None = None):
        """Reference two-stage semantics (`src/raglite/_search.py:66-79,143-149`; with chunk_filter the
        filter-first branch `:105-119`, with rank_limit the order-first-then-filter branch `:120-141`): returns
        (scores (B,k), chunk ordinals (B,k), counts (B,))."""
        a = _Args()
        p_q, B, single = self._queries(a, queries)
        o_s, p_s = a.out((B, k), np.float32)
        o_c, p_c = a.out((B, k), np.int32)
        o_n, p_n = a.out((B,), np.int32)
        p_f = self._filter(a, chunk_filter)
        self._prep(a)
        check(lib().rl_search_chunks_ranked(self._handle, p_q, B, num_hits, k, p_f, int(rank_limit or 0), p_s, p_c, p_n,
                                            a.mem, a.stream))
        return (o_s[0], o_c[0], o_n[0]) if single else (o_s, o_c, o_n)

    # -- a9 ----------------------------------------------------------------------------------------
    def maxsim_scores(self, query_vecs):
        a = _Args()
        p_q, nq, _ = self._queries(a, query_vecs)
        o_s, p_s = a.out((self.n_chunks,), np.float32)
        self._prep(a)
        check(lib().rl_maxsim_scores(self._handle, p_q, nq, p_s, a.mem, a.stream))
        return o_s

    def maxsim_topk(self, query_vecs, k: int, chunk_filter=None):
        a = _Args()
        p_q, nq, _ = self._queries(a, query_vecs)
        o_s, p_s = a.out((k,), np.float32)
        o_c, p_c = a.out((k,), np.int32)
        p_f = self._filter(a, chunk_filter)
        self._prep(a)
        check(lib().rl_maxsim_topk_filtered(self._handle, p_q, nq, k, p_f, p_s, p_c, a.mem, a.stream))
        return o_s, o_c

    def maxsim_topk_batch(self, query_batch, k: int):
        """query_batch (n_queries, nq, dim) -> (scores (n_queries, k), chunk ordinals (n_queries, k)); one corpus
        pass per query and one batched selection launch."""
        a = _Args()
        # fp16 queries -- what the reference's embed_strings / query adapter hand over (`_embed.py:140`, `_search.py:62`) -- go in as fp16
        # (`rl_maxsim_topk_batch_f16`): over an fp16-stored index the one-product pass is then exact and its top-k is the result
        half = _is_half(query_batch)
        p_q = a.inp(query_batch, np.float16 if half else np.float32)
        qv = a.keep[-1]
        if qv.ndim != 3 or int(qv.shape[2]) != self.dim:
            raise ValueError("query_batch must be (n_queries, nq, dim)")
        n_queries, nq = int(qv.shape[0]), int(qv.shape[1])
        o_s, p_s = a.out((n_queries, k), np.float32)
        o_c, p_c = a.out((n_queries, k), np.int32)
        self._prep(a)
        fn = lib().rl_maxsim_topk_batch_f16 if half else lib().rl_maxsim_topk_batch
        check(fn(self._handle, p_q, n_queries, nq, k, p_s, p_c, a.mem, a.stream))
        return o_s, o_c

    def maxsim_approx_scores(self, query_batch, kernel: int = 0):
        """The first stage of `maxsim_topk_batch`'s bound-filtered pipeline alone (`rl_maxsim_approx_scores`): approximate scores
        (n_queries, n_chunks) from the hi halves of corpus and queries, and per query the rigorous bound m with
        |approximate - exact| <= m for every chunk.  kernel 0: sixteen queries per pass (maxsim_pp.hip), 1: eight (maxsim_gemm.hip)."""
        a = _Args()
        p_q = a.inp(query_batch, np.float32)
        qv = a.keep[-1]
        if qv.ndim != 3 or int(qv.shape[2]) != self.dim:
            raise ValueError("query_batch must be (n_queries, nq, dim)")
        n_queries, nq = int(qv.shape[0]), int(qv.shape[1])
        o_s, p_s = a.out((n_queries, self.n_chunks), np.float32)
        o_b, p_b = a.out((n_queries,), np.float32)
        self._prep(a)
        check(lib().rl_maxsim_approx_scores(self._handle, p_q, n_queries, nq, int(kernel), p_s, p_b, a.mem, a.stream))
        return o_s, o_b

    def maxsim_batch_begin(self, query_batch, k: int):
        """First half of `maxsim_topk_batch` over one shard of a SHARDED corpus (`rl_maxsim_batch_begin`): the approximate passes; returns
        (n_queries, k + 1) float32 -- this shard's k best approximate scores per query (descending) and its error bound -- for the
        all-gather in front of `maxsim_batch_finish`.  Raises RaglitHipError(UNSUPPORTED) where the bound-filtered pipeline does not cover
        the batch: use `maxsim_topk_batch` then."""
        a = _Args()
        p_q = a.inp(query_batch, np.float32)
        qv = a.keep[-1]
        if qv.ndim != 3 or int(qv.shape[2]) != self.dim:
            raise ValueError("query_batch must be (n_queries, nq, dim)")
        n_queries, nq = int(qv.shape[0]), int(qv.shape[1])
        o, p_o = a.out((n_queries, int(k) + 1), np.float32)
        self._prep(a)
        check(lib().rl_maxsim_batch_begin(self._handle, p_q, n_queries, nq, int(k), p_o, a.mem, a.stream))
        return o

    def maxsim_batch_finish(self, query_batch, all_approx, rank: int, k: int):
        """Second half (`rl_maxsim_batch_finish`): all_approx (world, n_queries, k + 1) = every shard's `maxsim_batch_begin` result;
        returns (scores (n_queries, k), LOCAL chunk ordinals (n_queries, k)) of this shard's chunks that could be in the global top-k,
        ranked by exact score, padded with (-inf, -1)."""
        a = _Args()
        p_q = a.inp(query_batch, np.float32)
        qv = a.keep[-1]
        n_queries = int(qv.shape[0])
        p_a = a.inp(all_approx, np.float32)
        av = a.keep[-1]
        if av.ndim != 3 or int(av.shape[1]) != n_queries or int(av.shape[2]) != int(k) + 1:
            raise ValueError("all_approx must be (world, n_queries, k + 1)")
        world = int(av.shape[0])
        o_s, p_s = a.out((n_queries, int(k)), np.float32)
        o_c, p_c = a.out((n_queries, int(k)), np.int32)
        self._prep(a)
        check(lib().rl_maxsim_batch_finish(self._handle, p_q, p_a, world, int(rank), p_s, p_c, a.mem, a.stream))
        return o_s, o_c

    def maxsim_rerank(self, query_vecs, candidates):
        """query_vecs (n_queries, nq, dim), candidates (n_queries, n_cand) int32 -> scores (n_queries, n_cand)."""
        a = _Args()
        p_q = a.inp(query_vecs, np.float32)
        qv = a.keep[-1]
        if qv.ndim != 3 or int(qv.shape[2]) != self.dim:
            raise ValueError("query_vecs must be (n_queries, nq, dim)")
        p_c = a.inp(candidates, np.int32)
        cv = a.keep[-1]
        n_queries, nq = int(qv.shape[0]), int(qv.shape[1])
        if cv.ndim != 2 or int(cv.shape[0]) != n_queries:
            raise ValueError("candidates must be (n_queries, n_cand)")
        n_cand = int(cv.shape[1])
        o_s, p_s = a.out((n_queries, n_cand), np.float32)
        self._prep(a)
        check(lib().rl_maxsim_rerank(self._handle, p_q, n_queries, nq, p_c, n_cand, p_s, a.mem, a.stream))
        return o_s

    # -- 8f-3: device half of update_query_adapter ---------------------------------------------------
    def chunk_best_rows(self, queries, candidates):
        """For every (query b, chunk candidates[b][j]): the row ordinal maximising q . row, i.e.
        `np.argmax(chunk.embedding_matrix @ q)` (`src/raglite/_query_adapter.py:174,180`); -1 for candidate -1."""
        a = _Args()
        p_q, B, single = self._queries(a, queries)
        c2 = candidates.reshape(1, -1) if candidates.ndim == 1 else candidates
        p_c = a.inp(c2, np.int32)
        if int(a.keep[-1].shape[0]) != B:
            raise ValueError("candidates must have one row per query")
        n_cand = int(a.keep[-1].shape[1])
        o_r, p_r = a.out((B, n_cand), np.int32)
        self._prep(a)
        check(lib().rl_chunk_best_rows(self._handle, p_q, B, p_c, n_cand, p_r, a.mem, a.stream))
        return o_r[0] if single else o_r

    def gather_rows(self, rows):
        """Embedding rows as float32 (whatever the storage precision)."""
        a = _Args()
        p_r = a.inp(rows, np.int32)
        n = int(a.keep[0].shape[0])
        o, p_o = a.out((n, self.dim), np.float32)
        self._prep(a)
        check(lib().rl_gather_rows(self._handle, p_r, n, p_o, a.mem, a.stream))
        return o

    def time_kernel(self, kind: int, query_vecs_cuda, iters: int) -> float:
        """Milliseconds (HIP events on the launch stream) for `iters` launches of the dominant kernel."""
        a = _Args()
        p_q, nq, _ = self._queries(a, query_vecs_cuda)
        if a.mem != MEM_DEVICE:
            raise ValueError("time_kernel needs CUDA tensors")
        ms = C.c_float(0.0)
        self._prep(a)
        check(lib().rl_time_kernel(self._handle, kind, p_q, nq, iters, C.byref(ms), a.stream))
        return float(ms.value)
