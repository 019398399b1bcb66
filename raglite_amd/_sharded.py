"""Multi-GPU retrieval: corpus sharded by chunk, ONE all-gather of the local top-k over RCCL, merge.

SURVEY.md section 8e.  The reference is single-process; this is the only place a collective exists.
One process per GPU.  Each rank owns a contiguous range of chunks (all rows of a chunk on one rank, so the
per-chunk max of `src/raglite/_search.py:143-149` and MaxSim stay local), runs the single-GPU kernels over
its shard, and contributes `B x k x 8 B` (score bits, global id) to ONE all-gather; every rank then merges
`world x k` candidates per query.  At B = 1000, k = 100 that is 0.8 MB per rank -- microseconds over a
153 GB/s xGMI link next to a ~10 ms scan, so no bucketing / overlap machinery is warranted.

Two transports for that one step:
  * `comm=` a `raglite_amd.Communicator`: the exchange and the merge run behind the C ABI (`rl_allgather_merge_topk`:
    pack -> `ncclAllGather` through librccl -> `rl_merge_topk`'s kernel), CUDA tensors in and out, nothing
    synchronises with the host.  This is the production path; torch.distributed is not on it.
  * otherwise `torch.distributed` (`group=`): backend "nccl" (= RCCL) for CUDA tensors, "gloo" for the CPU
    tests; NumPy callers get the host merge below.
"""

from __future__ import annotations

from typing import Any

import numpy as np


SHARD_MISSING = -2  # include/raglite_hip.h: RL_ID_SHARD_MISSING -- the id a rank that failed in its local step sends into the merge


def shard_bounds_by_chunk(chunk_offsets, world: int) -> list[tuple[int, int]]:
    """Chunk ranges [lo, hi) per rank: contiguous, balanced by row count, chunks never split."""
    off = np.asarray(chunk_offsets, dtype=np.int64)
    n_chunks, n_rows = len(off) - 1, int(off[-1])
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(off, (n_rows * r) // world, side="left"))
        cuts.append(min(max(c, cuts[-1]), n_chunks))
    cuts.append(n_chunks)
    return list(zip(cuts[:-1], cuts[1:]))


def _merge_order(s: np.ndarray, i: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Positions of the k best entries of every row of s / i (shape (B, M); id -1 = padding) by (score desc, id asc),
    NaN after -inf, padding last; and how many of them are real.  One lexsort over the whole batch."""
    pad = i < 0
    nan = np.isnan(s)
    key = np.where(nan | pad, -np.inf, s)
    order = np.lexsort((i, -key, nan, pad), axis=-1)[:, :k]
    n_valid = np.minimum((~pad).sum(axis=1), k)
    return order, n_valid


def merge_topk_host(scores: np.ndarray, ids: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Global top-k of per-shard lists.  scores/ids: (world, B, k_in); ids are GLOBAL, -1 = padding.
    Order (score desc, id asc), NaN last -- identical to what one GPU holding everything returns."""
    world, B, k_in = scores.shape
    s = np.ascontiguousarray(np.transpose(scores, (1, 0, 2))).reshape(B, world * k_in)
    i = np.ascontiguousarray(np.transpose(ids, (1, 0, 2))).reshape(B, world * k_in).astype(np.int64)
    order, n_valid = _merge_order(s, i, k)
    out_s = np.full((B, k), -np.inf, dtype=np.float32)
    out_i = np.full((B, k), -1, dtype=np.int64)
    kk = order.shape[1]
    real = np.arange(kk)[None, :] < n_valid[:, None]
    out_s[:, :kk] = np.where(real, np.take_along_axis(s, order, axis=1), -np.inf)
    out_i[:, :kk] = np.where(real, np.take_along_axis(i, order, axis=1), -1)
    return out_s, out_i


def group_chunk_max_host(row_scores: np.ndarray, row_chunks: np.ndarray, k: int):
    """a8 on the host for the sharded two-stage search: hits are sorted (score desc, row asc), a hit is
    kept iff it is the first of its chunk (`src/raglite/_search.py:143-149`).  Array code over the whole batch."""
    row_scores = np.asarray(row_scores)
    row_chunks = np.asarray(row_chunks).astype(np.int64)
    B, H = row_chunks.shape
    out_s = np.full((B, k), -np.inf, dtype=np.float32)
    out_c = np.full((B, k), -1, dtype=np.int64)
    if H == 0:
        return out_s, out_c, np.zeros(B, dtype=np.int32)
    by_chunk = np.argsort(row_chunks, axis=1, kind="stable")  # within a chunk the hits keep their rank order
    sorted_chunks = np.take_along_axis(row_chunks, by_chunk, axis=1)
    first_sorted = np.ones((B, H), dtype=bool)
    first_sorted[:, 1:] = sorted_chunks[:, 1:] != sorted_chunks[:, :-1]
    first = np.zeros((B, H), dtype=bool)
    np.put_along_axis(first, by_chunk, first_sorted, axis=1)
    first &= row_chunks >= 0
    rank = np.cumsum(first, axis=1) - 1  # position of a kept hit among the kept hits of its query
    keep = first & (rank < k)
    b_idx, h_idx = np.nonzero(keep)
    out_s[b_idx, rank[b_idx, h_idx]] = row_scores[b_idx, h_idx]
    out_c[b_idx, rank[b_idx, h_idx]] = row_chunks[b_idx, h_idx]
    return out_s, out_c, keep.sum(axis=1).astype(np.int32)


def merge_order_torch(s, i, k: int):
    """`_merge_order` on torch tensors (any device, nothing synchronises with the host): three stable sorts, least significant
    key first -- id asc, score desc, then (padding, NaN) last."""
    import torch

    pad = i < 0
    nan = torch.isnan(s)
    key = torch.where(nan | pad, torch.full_like(s, float("-inf")), s)
    idx = torch.argsort(i, dim=1, stable=True)
    idx = idx.gather(1, torch.argsort(key.gather(1, idx), dim=1, descending=True, stable=True))
    cls = pad.to(torch.int32) * 2 + nan.to(torch.int32)
    idx = idx.gather(1, torch.argsort(cls.gather(1, idx), dim=1, stable=True))
    order = idx[:, :k]
    n_valid = torch.clamp((~pad).sum(dim=1), max=k)
    return order, n_valid


def group_chunk_max_torch(row_scores, row_chunks, k: int):
    """`group_chunk_max_host` on torch tensors (any device, no host synchronisation: kept hits are scattered to their rank, the
    rest to a spare column).  Returns (scores (B,k) float32, chunks (B,k) int64, counts (B,) int32)."""
    import torch

    B, H = row_chunks.shape
    dev = row_chunks.device
    out_s = torch.full((B, k + 1), float("-inf"), dtype=torch.float32, device=dev)
    out_c = torch.full((B, k + 1), -1, dtype=torch.int64, device=dev)
    if H == 0:
        return out_s[:, :k], out_c[:, :k], torch.zeros(B, dtype=torch.int32, device=dev)
    row_chunks = row_chunks.to(torch.int64)
    by_chunk = torch.argsort(row_chunks, dim=1, stable=True)  # within a chunk the hits keep their rank order
    sorted_chunks = row_chunks.gather(1, by_chunk)
    first_sorted = torch.ones((B, H), dtype=torch.bool, device=dev)
    first_sorted[:, 1:] = sorted_chunks[:, 1:] != sorted_chunks[:, :-1]
    first = torch.zeros((B, H), dtype=torch.bool, device=dev).scatter_(1, by_chunk, first_sorted)
    first &= row_chunks >= 0
    rank = torch.cumsum(first.to(torch.int64), dim=1) - 1
    keep = first & (rank < k)
    slot = torch.where(keep, rank, torch.full_like(rank, k))
    out_s.scatter_(1, slot, row_scores.to(torch.float32))
    out_c.scatter_(1, slot, row_chunks)
    return out_s[:, :k].contiguous(), out_c[:, :k].contiguous(), keep.sum(dim=1).to(torch.int32)


def _all_gather_stacked(t, group):
    """(world, *t.shape) tensor of every rank's `t`.  The output is allocated in the CONCATENATED form
    `(world * t.shape[0], ...)` -- the one every backend's `all_gather_into_tensor` accepts (gloo rejects the stacked
    form) -- and viewed as a stack afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    t = t.contiguous()
    flat = torch.empty((world * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(flat, t, group=group)
    return flat.view(world, *t.shape)


def _to_numpy(x: Any) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _is_cuda(x: Any) -> bool:
    return bool(getattr(x, "is_cuda", False))


class ShardedIndex:
    """This rank's shard plus the exchange step.

    local                 object with `search_rows(q, k)`, `maxsim_topk(Q, k)` returning LOCAL ordinals
                          (a `raglite_amd.DeviceIndex` over the shard's rows)
    row_base, chunk_base  global ordinal of the shard's first row / chunk
    local_chunk_offsets   the shard's CSR (needed by `search_chunks`)
    group                 torch.distributed process group (None = default group) for the torch transport
    comm                  a `raglite_amd.Communicator`: the exchange goes through the C ABI / librccl instead
    """

    def __init__(self, local: Any, *, row_base: int, chunk_base: int, local_chunk_offsets=None, group=None, comm=None,
                 check_failures: bool = False) -> None:
        """check_failures: device-path MaxSim batches read back one flag per call to learn that ANOTHER rank failed in its local step (a
        host synchronisation per call; host-array calls check for free).  Without it a failing rank still takes part in every collective
        of the call -- with empty lists whose ids are SHARD_MISSING, on EVERY route: the staged one and the ones that go straight to the
        merge (fp16 queries over fp16-stored shards, tiny batches) -- so nobody hangs, and raises afterwards; the other ranks do not
        raise, but what they return is POISONED on the device by the merge itself (`rl_allgather_merge_topk`: every score NaN, every id
        -1): a merge that lacks a shard never looks like an answer."""
        self.check_failures = bool(check_failures)
        self.local = local
        self.row_base = int(row_base)
        self.chunk_base = int(chunk_base)
        self.local_chunk_offsets = None if local_chunk_offsets is None else np.asarray(local_chunk_offsets, np.int64)
        self.group = group
        self.comm = comm

    # -- the one collective ------------------------------------------------------------------------
    def _world(self) -> int:
        if self.comm is not None:
            return self.comm.world
        import torch.distributed as dist

        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _all_gather(self, packed: np.ndarray) -> np.ndarray:
        """packed: int32 array (identical shape on every rank) -> (world, *shape), torch transport, host arrays."""
        import torch
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return packed[None]
        world = dist.get_world_size(self.group)
        backend = dist.get_backend(self.group)
        t = torch.from_numpy(np.ascontiguousarray(packed))
        if backend == "nccl":  # RCCL moves device memory
            t = t.cuda()
        try:
            out = _all_gather_stacked(t, self.group)
        except (RuntimeError, NotImplementedError):  # backends without all_gather_into_tensor
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t, group=self.group)
            out = torch.stack(parts)
        return out.cpu().numpy()

    def _allgather_host(self, x: np.ndarray) -> np.ndarray:
        """int32 host array (same shape on every rank) -> (world, *shape) through WHATEVER transport this index has: torch.distributed
        when it is initialised, else the attached Communicator (librccl through the C ABI, staged through a device tensor), else -- one
        rank -- the array itself.  Several ranks and no transport for host arrays is an error, not a silent "world of one"."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return self._all_gather(x)
        if self.comm is not None and self.comm.world > 1:
            import torch

            t = torch.from_numpy(x)
            if torch.cuda.is_available():  # (librccl moves device memory; a test double without a device takes the host tensor)
                t = t.cuda()
            return self.comm.allgather(t).cpu().numpy()
        if self._world() > 1:
            raise RuntimeError("ShardedIndex: several ranks but neither torch.distributed nor a Communicator can exchange host arrays")
        return x[None]

    @staticmethod
    def _pack(*cols: np.ndarray) -> np.ndarray:
        return np.stack([np.ascontiguousarray(c).view(np.int32) if c.dtype == np.float32 else c.astype(np.int32)
                         for c in cols], axis=-1)

    def _exchange_host(self, scores, ids_local, base: int, extra=None):
        """Host arrays: pack (score bits, global id[, extra]) -> ONE all-gather -> (world, B, k) arrays."""
        s = _to_numpy(scores).astype(np.float32, copy=False)
        i = _to_numpy(ids_local).astype(np.int64)
        s2 = s.reshape(1, -1) if s.ndim == 1 else s
        i2 = i.reshape(1, -1) if i.ndim == 1 else i
        gid = np.where(i2 >= 0, i2 + base, i2)  # (-1: an empty slot; SHARD_MISSING: this rank failed in its local step)
        cols = [s2, gid] if extra is None else [s2, gid, extra]
        # (through WHATEVER transport the index has: a Communicator-only index without torch.distributed used to merge its own shard alone here)
        g = self._allgather_host(self._pack(*cols))  # (world, B, k, ncols)
        gs = np.ascontiguousarray(g[..., 0]).view(np.float32)
        return gs, g[..., 1], (g[..., 2] if extra is not None else None), s.ndim == 1

    def _exchange_merge_device(self, scores, ids_local, base: int, k: int):
        """CUDA tensors in, CUDA tensors out, no host synchronisation: (score bits, global id) records, ONE all-gather
        (RCCL), merge with `rl_merge_topk`'s kernel.  Lets consecutive query batches queue back to back on the stream
        (the host only enqueues).  Through the C ABI when a Communicator is attached."""
        import torch

        big = self._world() * int(scores.shape[-1]) > 8192  # more candidates per query than rl_merge_topk's kernel sorts in LDS
        if self.comm is not None and not big:
            # pack (ids made global, markers kept) -> ONE ncclAllGather -> merge -> poisoned when a shard is missing: all behind the C ABI
            return self.comm.allgather_merge_topk(scores, ids_local.to(torch.int32), base, k)
        from . import _ops

        if self.comm is not None:
            gs, gi = self.comm.allgather_topk(scores, ids_local.to(torch.int32), base)
        else:
            if self._world() == 1 and base == 0:  # one shard that starts at ordinal 0: local ordinals ARE the global ones (no kernel at all)
                return scores, ids_local.to(torch.int32)
            gid = torch.where(ids_local >= 0, ids_local + base, ids_local).to(torch.int32)
            if self._world() == 1:
                return scores, gid
            packed = torch.stack([scores.contiguous().view(torch.int32), gid], dim=-1).contiguous()  # (B, k, 2)
            out = _all_gather_stacked(packed, self.group)
            gs = out[..., 0].contiguous().view(torch.float32)  # (world, B, k)
            gi = out[..., 1].contiguous()
        # (the torch transport / the > 8192-record merge: what rl_allgather_merge_topk does behind the C ABI, as torch kernels)
        missing = (gi == SHARD_MISSING).any()
        if not big:
            ms, mi = _ops.merge_topk(gs, gi, k)
        else:
            # e.g. k = 2048 on 8 GPUs: the same ordering by three stable sorts on the device (merge_order_torch), still no host sync
            world, B, kin = gs.shape
            fs = gs.permute(1, 0, 2).reshape(B, world * kin)
            fi = gi.permute(1, 0, 2).reshape(B, world * kin).to(torch.int64)
            order, n_valid = merge_order_torch(fs, fi, k)
            real = torch.arange(order.shape[1], device=fs.device)[None, :] < n_valid[:, None]
            ms = torch.where(real, fs.gather(1, order), torch.full((), float("-inf"), device=fs.device))
            mi = torch.where(real, fi.gather(1, order), torch.full((), -1, dtype=torch.int64, device=fs.device)).to(torch.int32)
        return torch.where(missing, torch.full_like(ms, float("nan")), ms), torch.where(missing, torch.full_like(mi, -1), mi)

    def _gather_device(self, scores, ids, offset: int):
        """(B, k) CUDA lists -> (world, B, k) of every rank's, ids + offset (-1 stays -1)."""
        import torch

        if self.comm is not None:
            return self.comm.allgather_topk(scores, ids.to(torch.int32), offset)
        gid = torch.where(ids >= 0, ids + offset, ids).to(torch.int32)
        if self._world() == 1:
            return scores[None], gid[None]
        packed = torch.stack([scores.contiguous().view(torch.int32), gid], dim=-1).contiguous()
        out = _all_gather_stacked(packed, self.group)
        return out[..., 0].contiguous().view(torch.float32), out[..., 1].contiguous()

    # -- filters ---------------------------------------------------------------------------------------------
    def _local_filter(self, chunk_filter):
        """`chunk_filter`: bool mask over the GLOBAL chunk ordinals (the host's evaluation of `metadata_filter`,
        `src/raglite/_search.py:84-97`) -> this shard's slice of it (None stays None)."""
        if chunk_filter is None:
            return None
        n_local = (len(self.local_chunk_offsets) - 1) if self.local_chunk_offsets is not None else int(self.local.n_chunks)
        mask = np.asarray(_to_numpy(chunk_filter), dtype=bool)
        if mask.ndim != 1 or len(mask) < self.chunk_base + n_local:
            raise ValueError("chunk_filter must be a bool mask over all (global) chunk ordinals")
        return np.ascontiguousarray(mask[self.chunk_base : self.chunk_base + n_local])

    @staticmethod
    def _kw(chunk_filter=None, rank_limit=None) -> dict:
        kw = {}
        if chunk_filter is not None:
            kw["chunk_filter"] = chunk_filter
        if rank_limit:
            kw["rank_limit"] = int(rank_limit)
        return kw

    # -- searches ----------------------------------------------------------------------------------------
    # -- small integer collectives of the global rank cut (host arrays: torch.distributed; CUDA tensors: the Communicator or RCCL) ----
    def _allreduce_sum_int(self, x):
        if self._world() == 1:
            return x
        if _is_cuda(x):
            if self.comm is not None:
                return self.comm.allreduce_sum_(x.contiguous())
            import torch.distributed as dist

            x = x.contiguous()
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            return x
        return self._allgather_host(x).astype(np.int64).sum(axis=0).astype(np.int32)

    def _allgather_int(self, x):
        """(world, *x.shape)."""
        if _is_cuda(x):
            if self.comm is not None:
                return self.comm.allgather(x)
            return x[None] if self._world() == 1 else _all_gather_stacked(x, self.group)
        return self._allgather_host(x)

    def _rank(self) -> int:
        if self.comm is not None:
            return self.comm.rank
        import torch.distributed as dist

        return dist.get_rank(self.group) if (dist.is_available() and dist.is_initialized()) else 0

    def _local_rows_ranked(self, queries, k: int, chunk_filter, rank_limit):
        """The local search of `search_rows` / `search_chunks`.  With a rank cut over several shards the cut is the GLOBAL one
        (`ORDER BY dist LIMIT rank_limit` over the whole table, `_search.py:120-141`): the shards walk the three radix levels of the
        cut together -- each level's histogram summed over the ranks -- and take ties on the threshold in global row order
        (`rl_rank_cut_*`); shards hold ascending row ranges in rank order.  Falls back to the per-shard cut (a superset) for local
        objects without the staged calls."""
        local_filter = self._local_filter(chunk_filter)
        if not rank_limit or self._world() == 1 or not hasattr(self.local, "rank_cut_begin"):
            return self.local.search_rows(queries, k, **self._kw(local_filter, rank_limit))
        # every rank's row count, through the transport this index really has (with a Communicator and no torch.distributed the torch
        # transport would report a world of one: the cut would be decided per rank, and the ranks would take different branches)
        counts = self._allgather_host(np.asarray([int(self.local.n_rows)], dtype=np.int32)).astype(np.int64).reshape(-1)
        n_total, n_max = int(counts.sum()), int(counts.max())
        if int(rank_limit) >= n_total:  # no cut at all: the filter-first search
            return self.local.search_rows(queries, k, **self._kw(local_filter, None))
        single = getattr(queries, "ndim", 2) == 1
        # `rl_rank_cut_begin` scores the whole batch into one [B x n_local] matrix (<= 8 GB).  The sub-batch size is derived from the
        # LARGEST shard, so that every rank makes the same split and enters the same collectives the same number of times.
        B = 1 if single else int(queries.shape[0])
        per = max(1, int(self.rank_cut_scratch_bytes // (4 * max(n_max, 1) + 64)))
        if B <= per:
            s, r = self._rank_cut_batch(queries, k, local_filter, rank_limit)
            return (s[0], r[0]) if single else (s, r)
        outs = [self._rank_cut_batch(queries[b0 : b0 + per], k, local_filter, rank_limit) for b0 in range(0, B, per)]
        if _is_cuda(outs[0][0]):
            import torch

            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        return np.concatenate([_to_numpy(o[0]) for o in outs]), np.concatenate([_to_numpy(o[1]) for o in outs])

    def _rank_cut_batch(self, queries, k: int, local_filter, rank_limit):
        """One (sub-)batch through the staged global rank cut; (scores (B, k), local rows (B, k))."""
        self.local.rank_cut_begin(queries)
        for level in range(3):
            self.local.rank_cut_level_done(level, self._allreduce_sum_int(self.local.rank_cut_level(level, rank_limit)))
        ties = self.local.rank_cut_ties(rank_limit)
        all_ties = self._allgather_int(ties)  # (world, B)
        before = all_ties[: self._rank()].sum(0) if self._rank() > 0 else all_ties[0] * 0
        if _is_cuda(ties):
            import torch

            before = before.to(torch.int32)
        else:
            before = np.asarray(before, dtype=np.int32)
        return self.local.rank_cut_finish(rank_limit, before, k, chunk_filter=local_filter)

    def search_rows(self, queries, k: int, chunk_filter=None, rank_limit: int | None = None):
        """Global exact top-k rows: (scores (B,k), global row ordinals (B,k)).  chunk_filter: bool mask over the GLOBAL chunk
        ordinals (filter-first branch, `_search.py:105-119`).  rank_limit: the order-first branch's cut (`:120-141`), over the WHOLE
        corpus (see `_local_rows_ranked`): the same rows, bit for bit, as one index holding everything returns."""
        s, r = self._local_rows_ranked(queries, k, chunk_filter, rank_limit)
        if _is_cuda(s):  # device-resident queries (cfg 5: B = 1000): merge on the device too
            single = s.dim() == 1
            ms, mi = self._exchange_merge_device(s.reshape(1, -1) if single else s, r.reshape(1, -1) if single else r,
                                                 self.row_base, k)
            return (ms[0], mi[0]) if single else (ms, mi)
        gs, gi, _, single = self._exchange_host(s, r, self.row_base)
        ms, mi = merge_topk_host(gs, gi, k)
        return (ms[0], mi[0]) if single else (ms, mi)

    def maxsim_topk(self, query_vecs, k: int, chunk_filter=None):
        """Global exact top-k chunks by MaxSim: (scores (k,), global chunk ordinals (k,)); chunk_filter as in `search_rows`."""
        s, c = self.local.maxsim_topk(query_vecs, k, **self._kw(self._local_filter(chunk_filter)))
        if _is_cuda(s):
            ms, mi = self._exchange_merge_device(s.reshape(1, -1), c.reshape(1, -1), self.chunk_base, k)
            return ms[0], mi[0]
        gs, gi, _, _ = self._exchange_host(s, c, self.chunk_base)
        ms, mi = merge_topk_host(gs, gi, k)
        return ms[0], mi[0]

    def maxsim_topk_batch(self, query_batch, k: int):
        """A batch of queries (QB, nq, dim): the local batched search, ONE all-gather of (QB, k, 2) int32, one merge
        (on the device without any host synchronisation when the queries are CUDA tensors, else on the host).
        Returns (scores (QB,k), global chunk ordinals (QB,k))."""
        self._deferred_error = None
        if hasattr(self.local, "maxsim_topk_batch"):
            s, c = self._local_maxsim_batch(query_batch, k)
            if _is_cuda(s):  # device-resident queries: results stay on the device
                # a rank that failed in its local step sent SHARD_MISSING ids: the merge every rank gets back is poisoned on the device
                # (every score NaN, every id -1 -- `rl_allgather_merge_topk`), unusable rather than plausible, without a host read-back
                out = self._exchange_merge_device(s, c, self.chunk_base, k)
                if self.check_failures and self._deferred_error is None and self._world() > 1:
                    import torch

                    if bool(torch.isnan(out[0]).all().item()):
                        self._deferred_error = RuntimeError("ShardedIndex.maxsim_topk_batch: another rank failed in its local step; this merge lacks its shard")
                self._raise_deferred()
                return out
        else:
            outs = [self.local.maxsim_topk(query_batch[b], k) for b in range(len(query_batch))]
            s, c = np.stack([_to_numpy(o[0]) for o in outs]), np.stack([_to_numpy(o[1]) for o in outs])
        gs, gi, _, _ = self._exchange_host(s, c, self.chunk_base)
        if self._deferred_error is None and bool((gi == SHARD_MISSING).any()):
            self._deferred_error = RuntimeError("ShardedIndex.maxsim_topk_batch: another rank failed in its local step; this merge lacks its shard")
        self._raise_deferred()
        return merge_topk_host(gs, gi, k)

    _deferred_error = None
    rank_cut_scratch_bytes = 8 << 30  # what rl_rank_cut_begin accepts for its [B x n_local] score matrix

    def _raise_deferred(self) -> None:
        err, self._deferred_error = self._deferred_error, None
        if err is not None:
            raise err

    def _local_maxsim_batch(self, query_batch, k: int):
        """This shard's part of a MaxSim batch.  Over several shards the bound-filtered pipeline takes ONE candidate threshold for all of
        them (`rl_maxsim_batch_begin` -> all-gather of every shard's k best approximate scores and bound, (world, B, k + 1) float32 ->
        `rl_maxsim_batch_finish`): a shard on its own re-scores the chunks near ITS k-th best approximate score, ~235 per query on each of
        eight shards where one index re-scores 307 in all.  The decision to exchange is taken from the batch shape alone (the same on every
        rank: the exchange is a collective); a shard whose index cannot take part (no image of the hi halves) contributes an empty list
        and a zero bound -- which only lowers the others' threshold -- and answers with its exact local top-k."""
        # fp16 queries over fp16-STORED shards (what RAGLite's embeddings are on both sides, `_embed.py:140`): every shard's one-product pass
        # is exact (`rl_maxsim_topk_batch_f16`), its local top-k needs no candidate threshold -- nothing to exchange before the merge.  Taken
        # from the query dtype and the local storage, which are the same on every rank of one index (the exchange below is a collective).
        from ._ops import _is_half

        f16_exact = _is_half(query_batch) and getattr(self.local, "storage", None) == "f16"  # (IEEE fp16 exactly: bfloat16 is not it)
        staged_shape = (
            self._world() > 1
            and not f16_exact
            and hasattr(self.local, "maxsim_batch_begin")
            and getattr(query_batch, "ndim", 0) == 3
            and int(query_batch.shape[0]) >= 3
            and int(query_batch.shape[0]) % 8 not in (1, 2)
            and int(query_batch.shape[1]) <= 32
            and int(k) <= 512
        )
        B = int(query_batch.shape[0]) if getattr(query_batch, "ndim", 0) == 3 else len(query_batch)

        def empty_lists():
            """What a rank that failed locally feeds into the merge every other rank is about to enter: no candidates, ids SHARD_MISSING."""
            if _is_cuda(query_batch):
                import torch

                return (torch.full((B, int(k)), float("-inf"), dtype=torch.float32, device=query_batch.device),
                        torch.full((B, int(k)), SHARD_MISSING, dtype=torch.int32, device=query_batch.device))
            return np.full((B, int(k)), -np.inf, dtype=np.float32), np.full((B, int(k)), SHARD_MISSING, dtype=np.int32)

        if not staged_shape:
            if self._world() == 1:
                return self.local.maxsim_topk_batch(query_batch, k)
            # several ranks, no threshold exchange (fp16 queries over fp16-stored shards, tiny batches, k > 512): the merge all-gather is
            # still a collective the other ranks enter -- a local failure (out of memory, a HIP error, a first-call image build) must
            # not leave them waiting in it
            try:
                return self.local.maxsim_topk_batch(query_batch, k)
            except Exception as exc:  # noqa: BLE001
                self._deferred_error = exc
                return empty_lists()
        from ._abi import UnsupportedError

        failed = None  # an exception that is NOT "this index cannot take part": this rank still enters every collective, then raises
        try:
            approx = self.local.maxsim_batch_begin(query_batch, k)
            staged = True
        except Exception as exc:  # noqa: BLE001 - whatever it is, the other ranks are already waiting in the all-gather
            staged = False
            if not isinstance(exc, UnsupportedError):
                failed = exc
            if _is_cuda(query_batch):
                import torch

                approx = torch.full((B, int(k) + 1), float("-inf"), dtype=torch.float32, device=query_batch.device)
            else:
                approx = np.full((B, int(k) + 1), -np.inf, dtype=np.float32)
            approx[:, int(k)] = float("nan") if failed is not None else 0.0  # (a NaN bound marks the failure; fmaxf ignores it on the device)
        if _is_cuda(approx):
            import torch

            all_approx = self._allgather_int(approx.contiguous().view(torch.int32)).view(torch.float32)
        else:
            all_approx = np.ascontiguousarray(self._allgather_int(np.ascontiguousarray(approx, dtype=np.float32).view(np.int32))).view(np.float32)
        if failed is not None:  # empty lists into the merge that follows (nobody hangs, everybody's merge is poisoned), the error once it is through
            self._deferred_error = failed
            return empty_lists()
        if not staged:
            try:
                return self.local.maxsim_topk_batch(query_batch, k)
            except Exception as exc:  # noqa: BLE001
                self._deferred_error = exc
                return empty_lists()
        try:
            return self.local.maxsim_batch_finish(query_batch, all_approx, self._rank(), k)
        except Exception as exc:  # noqa: BLE001 - one more collective (the merge) is still ahead of every rank
            self._deferred_error = exc
            return empty_lists()

    def search_chunks(self, queries, num_hits: int, k: int, chunk_filter=None, rank_limit: int | None = None):
        """Reference two-stage semantics across shards (`src/raglite/_search.py:66-79,143-149`; chunk_filter / rank_limit as in
        `search_rows`: the filtered branches `:105-141`): every rank's
        top-`num_hits` rows travel with their global chunk ordinals, are merged to the global top-`num_hits` rows
        (score desc, row asc), then grouped by chunk.  CUDA queries: two tiny all-gathers on the device (rows, chunks), then
        the merge + group-by as stable sorts and scatters on the device (`merge_order_torch`, `group_chunk_max_torch`): no host
        synchronisation, CUDA tensors back."""
        if self.local_chunk_offsets is None:
            raise ValueError("search_chunks needs local_chunk_offsets")
        s, r = self._local_rows_ranked(queries, num_hits, chunk_filter, rank_limit)
        device = s.device if _is_cuda(s) else None
        if device is not None:
            import torch

            single = s.dim() == 1
            s2, r2 = (s.reshape(1, -1), r.reshape(1, -1)) if single else (s, r)
            loc = torch.as_tensor(self.local_chunk_offsets, device=device)
            chunk_local = torch.searchsorted(loc, r2.to(torch.int64).clamp(min=0), right=True) - 1
            chunk_local = torch.where(r2 >= 0, chunk_local, torch.full_like(chunk_local, -1)).to(torch.int32)
            gs_t, gi_t = self._gather_device(s2, r2, self.row_base)
            _, gc_t = self._gather_device(s2, chunk_local, self.chunk_base)
            # merge + group-by on the device, nothing synchronises with the host: the same orderings as the host code below
            world, B, kin = gs_t.shape
            fs = gs_t.permute(1, 0, 2).reshape(B, world * kin)
            fr = gi_t.permute(1, 0, 2).reshape(B, world * kin).to(torch.int64)
            fc = gc_t.permute(1, 0, 2).reshape(B, world * kin).to(torch.int64)
            order, n_valid = merge_order_torch(fs, fr, num_hits)  # the global top-num_hits rows of every query
            real = torch.arange(order.shape[1], device=device)[None, :] < n_valid[:, None]
            ms = torch.where(real, fs.gather(1, order), torch.full((), float("-inf"), device=device))
            mc = torch.where(real, fc.gather(1, order), torch.full((), -1, dtype=torch.int64, device=device))
            o_s, o_c, o_n = group_chunk_max_torch(ms, mc, k)
            out = (o_s, o_c.to(torch.int32), o_n)
            return tuple(o[0] for o in out) if single else out
        else:
            r_np = _to_numpy(r).astype(np.int64)
            r2n = r_np.reshape(1, -1) if r_np.ndim == 1 else r_np
            chunk_local = np.searchsorted(self.local_chunk_offsets, r2n, side="right") - 1
            chunk_global = np.where(r2n >= 0, chunk_local + self.chunk_base, -1)
            gs, gi, gc, single = self._exchange_host(s, r, self.row_base, extra=chunk_global)
        world, B, kin = gs.shape
        fs = np.ascontiguousarray(np.transpose(gs, (1, 0, 2))).reshape(B, world * kin)
        fr = np.ascontiguousarray(np.transpose(gi, (1, 0, 2))).reshape(B, world * kin).astype(np.int64)
        fc = np.ascontiguousarray(np.transpose(gc, (1, 0, 2))).reshape(B, world * kin).astype(np.int64)
        order, n_valid = _merge_order(fs, fr, num_hits)  # the global top-num_hits rows of every query
        real = np.arange(order.shape[1])[None, :] < n_valid[:, None]
        ms = np.where(real, np.take_along_axis(fs, order, axis=1), -np.inf).astype(np.float32)
        mc = np.where(real, np.take_along_axis(fc, order, axis=1), -1)
        out = group_chunk_max_host(ms, mc, k)
        return tuple(o[0] for o in out) if single else out
