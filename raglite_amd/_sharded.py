"""Multi-GPU retrieval: corpus sharded by chunk, one RCCL all-gather of the local top-k, host merge.

SURVEY.md section 8e.  The reference is single-process; this is the only place a collective exists.
One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for
tests).  Each rank owns a contiguous range of chunks (all rows of a chunk on one rank, so the
per-chunk max of `src/raglite/_search.py:143-149` and MaxSim stay local), runs the single-GPU
kernels over its shard, and contributes `B x k x 8 B` (score bits, global id) to ONE
`all_gather_into_tensor`; every rank then merges `world x k` candidates per query on the host.
At B = 1000, k = 100 that is 0.8 MB per rank -- microseconds over a 153 GB/s xGMI link next to a
~20 ms scan, so no bucketing / overlap machinery is warranted.
"""

from __future__ import annotations

from typing import Any

import numpy as np


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


def merge_topk_host(scores: np.ndarray, ids: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Global top-k of per-shard lists.  scores/ids: (world, B, k_in); ids are GLOBAL, -1 = padding.
    Order (score desc, id asc), NaN last -- identical to what one GPU holding everything returns."""
    world, B, k_in = scores.shape
    s = np.transpose(scores, (1, 0, 2)).reshape(B, world * k_in)
    i = np.transpose(ids, (1, 0, 2)).reshape(B, world * k_in).astype(np.int64)
    out_s = np.full((B, k), -np.inf, dtype=np.float32)
    out_i = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        valid = i[b] >= 0
        sb, ib = s[b][valid], i[b][valid]
        nan = np.isnan(sb)
        order = np.lexsort((ib, -np.where(nan, -np.inf, sb), nan))[:k]
        out_s[b, : len(order)] = sb[order]
        out_i[b, : len(order)] = ib[order]
    return out_s, out_i


def group_chunk_max_host(row_scores: np.ndarray, row_chunks: np.ndarray, k: int):
    """a8 on the host for the sharded two-stage search: hits are sorted (score desc, row asc), a hit is
    kept iff it is the first of its chunk (`src/raglite/_search.py:143-149`)."""
    B = row_scores.shape[0]
    out_s = np.full((B, k), -np.inf, dtype=np.float32)
    out_c = np.full((B, k), -1, dtype=np.int64)
    counts = np.zeros(B, dtype=np.int32)
    for b in range(B):
        seen: set[int] = set()
        n = 0
        for s, c in zip(row_scores[b], row_chunks[b]):
            c = int(c)
            if c < 0 or c in seen:
                continue
            seen.add(c)
            if n < k:
                out_s[b, n], out_c[b, n] = s, c
                n += 1
        counts[b] = n
    return out_s, out_c, counts


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


class ShardedIndex:
    """This rank's shard plus the exchange step.

    local       object with `search_rows(q, k)`, `maxsim_topk(Q, k)` returning LOCAL ordinals
                (a `raglite_amd.DeviceIndex` over the shard's rows)
    row_base    global ordinal of the shard's first row;  chunk_base likewise for chunks
    row_chunks  optional callable local_rows -> local chunk ordinals (needed by `search_chunks`)
    group       torch.distributed process group (None = default group)
    """

    def __init__(self, local: Any, *, row_base: int, chunk_base: int, local_chunk_offsets=None, group=None) -> None:
        self.local = local
        self.row_base = int(row_base)
        self.chunk_base = int(chunk_base)
        self.local_chunk_offsets = None if local_chunk_offsets is None else np.asarray(local_chunk_offsets, np.int64)
        self.group = group

    # -- the one collective ------------------------------------------------------------------------
    def _all_gather(self, packed: np.ndarray) -> np.ndarray:
        """packed: int32 array (identical shape on every rank) -> (world, *shape)."""
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

    @staticmethod
    def _pack(*cols: np.ndarray) -> np.ndarray:
        return np.stack([np.ascontiguousarray(c).view(np.int32) if c.dtype == np.float32 else c.astype(np.int32)
                         for c in cols], axis=-1)

    def _exchange(self, scores, ids_local, base: int, extra=None):
        if hasattr(scores, "is_cuda") and scores.is_cuda:
            return self._exchange_device(scores, ids_local, base)
        s = _to_numpy(scores).astype(np.float32, copy=False)
        i = _to_numpy(ids_local).astype(np.int64)
        s2 = s.reshape(1, -1) if s.ndim == 1 else s
        i2 = i.reshape(1, -1) if i.ndim == 1 else i
        gid = np.where(i2 >= 0, i2 + base, -1)
        cols = [s2, gid] if extra is None else [s2, gid, extra]
        g = self._all_gather(self._pack(*cols))  # (world, B, k, ncols)
        gs = np.ascontiguousarray(g[..., 0]).view(np.float32)
        return gs, g[..., 1], (g[..., 2] if extra is not None else None), s.ndim == 1

    def _exchange_device(self, scores, ids_local, base: int):
        """CUDA tensors in: pack (score bits, global id) on the device, ONE all-gather (RCCL), one D2H."""
        import torch
        import torch.distributed as dist

        single = scores.dim() == 1
        s2 = scores.reshape(1, -1) if single else scores
        i2 = ids_local.reshape(1, -1) if single else ids_local
        gid = torch.where(i2 >= 0, i2 + base, torch.full_like(i2, -1))
        packed = torch.stack([s2.contiguous().view(torch.int32), gid.to(torch.int32)], dim=-1).contiguous()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            out = _all_gather_stacked(packed, self.group)
        else:
            out = packed[None]
        g = out.cpu().numpy()
        gs = np.ascontiguousarray(g[..., 0]).view(np.float32)
        return gs, g[..., 1], None, single

    def _exchange_merge_device(self, scores, ids_local, base: int, k: int):
        """CUDA tensors in, CUDA tensors out, no host synchronisation: pack (score bits, global id) on the device,
        ONE all-gather (RCCL), unpack, merge with `rl_merge_topk`.  Lets consecutive query batches queue
        back to back on the stream (the host only enqueues)."""
        import torch
        import torch.distributed as dist

        from . import _ops

        gid = torch.where(ids_local >= 0, ids_local + base, torch.full_like(ids_local, -1)).to(torch.int32)
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if world == 1:
            return scores, gid
        packed = torch.stack([scores.contiguous().view(torch.int32), gid], dim=-1).contiguous()  # (B, k, 2)
        out = _all_gather_stacked(packed, self.group)
        gs = out[..., 0].contiguous().view(torch.float32)  # (world, B, k)
        gi = out[..., 1].contiguous()
        return _ops.merge_topk(gs, gi, k)

    # -- searches ----------------------------------------------------------------------------------------
    def search_rows(self, queries, k: int):
        """Global exact top-k rows: (scores (B,k), global row ordinals (B,k))."""
        s, r = self.local.search_rows(queries, k)
        if hasattr(s, "is_cuda") and s.is_cuda:  # device-resident queries (cfg 5: B = 1000): merge on the device too
            single = s.dim() == 1
            ms, mi = self._exchange_merge_device(s.reshape(1, -1) if single else s, r.reshape(1, -1) if single else r,
                                                 self.row_base, k)
            return (ms[0], mi[0]) if single else (ms, mi)
        gs, gi, _, single = self._exchange(s, r, self.row_base)
        ms, mi = merge_topk_host(gs, gi, k)
        return (ms[0], mi[0]) if single else (ms, mi)

    def maxsim_topk(self, query_vecs, k: int):
        """Global exact top-k chunks by MaxSim: (scores (k,), global chunk ordinals (k,))."""
        s, c = self.local.maxsim_topk(query_vecs, k)
        if hasattr(s, "is_cuda") and s.is_cuda:
            ms, mi = self._exchange_merge_device(s.reshape(1, -1), c.reshape(1, -1), self.chunk_base, k)
            return ms[0], mi[0]
        gs, gi, _, _ = self._exchange(s, c, self.chunk_base)
        ms, mi = merge_topk_host(gs, gi, k)
        return ms[0], mi[0]

    def maxsim_topk_batch(self, query_batch, k: int):
        """A batch of queries (QB, nq, dim): QB local launches, ONE all-gather of (QB, k, 2) int32, one merge
        (on the device without any host synchronisation when the queries are CUDA tensors, else on the host).
        Returns (scores (QB,k), global chunk ordinals (QB,k))."""
        if hasattr(self.local, "maxsim_topk_batch"):
            s, c = self.local.maxsim_topk_batch(query_batch, k)
            if hasattr(s, "is_cuda") and s.is_cuda:  # device-resident queries: results stay on the device
                return self._exchange_merge_device(s, c, self.chunk_base, k)
        else:
            outs = [self.local.maxsim_topk(query_batch[b], k) for b in range(len(query_batch))]
            s, c = np.stack([_to_numpy(o[0]) for o in outs]), np.stack([_to_numpy(o[1]) for o in outs])
        gs, gi, _, _ = self._exchange(s, c, self.chunk_base)
        return merge_topk_host(gs, gi, k)

    def search_chunks(self, queries, num_hits: int, k: int):
        """Reference two-stage semantics across shards: gather each rank's top-`num_hits` rows with their
        global chunk ordinals, merge to the global top-`num_hits` rows, then group on the host."""
        if self.local_chunk_offsets is None:
            raise ValueError("search_chunks needs local_chunk_offsets")
        s, r = self.local.search_rows(queries, num_hits)
        r_np = _to_numpy(r).astype(np.int64)
        r2 = r_np.reshape(1, -1) if r_np.ndim == 1 else r_np
        chunk_local = np.searchsorted(self.local_chunk_offsets, r2, side="right") - 1
        chunk_global = np.where(r2 >= 0, chunk_local + self.chunk_base, -1)
        gs, gi, gc, single = self._exchange(s, r, self.row_base, extra=chunk_global)
        world, B, kin = gs.shape
        ms, mi = merge_topk_host(gs, gi, num_hits)
        # chunk ordinal of every merged row: look it up among the gathered (row, chunk) pairs
        flat_rows = np.transpose(gi, (1, 0, 2)).reshape(B, world * kin)
        flat_chunks = np.transpose(gc, (1, 0, 2)).reshape(B, world * kin)
        mc = np.full_like(mi, -1)
        for b in range(B):
            lut = {int(rr): int(cc) for rr, cc in zip(flat_rows[b], flat_chunks[b]) if rr >= 0}
            mc[b] = [lut.get(int(rr), -1) for rr in mi[b]]
        out = group_chunk_max_host(ms, mc, k)
        return tuple(o[0] for o in out) if single else out
