"""N>1 path on CPU: two processes over gloo run ShardedIndex end to end (shard bounds, global ids,
the single all-gather, host merge) with an oracle-backed local searcher standing in for the GPU
kernels, and must reproduce the single-shard oracle result exactly."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from raglite_amd._sharded import ShardedIndex, shard_bounds_by_chunk
from tests.util import ragged_offsets


class _OracleLocal:
    """Test double for raglite_amd.DeviceIndex over one shard (local ordinals, fp32 as-computed)."""

    def __init__(self, E, off, metric):
        self.E, self.off, self.metric = E, off, metric

    def search_rows(self, q, k, chunk_filter=None, rank_limit=None):
        assert rank_limit is None  # (over several shards the cut goes through the staged calls below)
        q2 = np.atleast_2d(q)
        S = np.full((len(q2), k), -np.inf, np.float32)
        I = np.full((len(q2), k), -1, np.int32)
        r2c = np.repeat(np.arange(len(self.off) - 1), np.diff(self.off))
        for b, qq in enumerate(q2):
            if chunk_filter is None:
                s, i = oracle.search_rows(self.E, qq, k, self.metric, np.float64)
            else:
                assert len(chunk_filter) == len(self.off) - 1  # the shard's slice of the global mask
                s, i = oracle.search_rows_filtered(self.E, r2c, qq, k, chunk_filter, self.metric, np.float64)
            S[b, : len(s)] = s
            I[b, : len(i)] = i
        return (S[0], I[0]) if np.ndim(q) == 1 else (S, I)

    # -- the staged rank cut (rl_rank_cut_*), restated in NumPy: order-preserving 32-bit keys, three radix levels (11 + 11 + 10 bits)
    @property
    def n_rows(self):
        return len(self.E)

    @staticmethod
    def _key(s):
        u = np.ascontiguousarray(s, dtype=np.float32).view(np.uint32).astype(np.uint64)
        key = np.where(u & 0x80000000, ~u & 0xFFFFFFFF, u | 0x80000000)
        return np.where(np.isnan(s), 0, key).astype(np.uint64)

    def search_rows_ranked_single(self, q, k, chunk_ok, rank_limit):
        r2c = np.repeat(np.arange(len(self.off) - 1), np.diff(self.off))
        return oracle.search_rows_ranked(self.E, r2c, q, k, chunk_ok, rank_limit, None, self.metric, np.float32)

    def rank_cut_begin(self, queries):
        q2 = np.atleast_2d(queries)
        self._sims = np.stack([oracle.similarity(self.E, qq, self.metric, np.float32).astype(np.float32) for qq in q2])
        self._keys = self._key(self._sims)
        self._hist = np.zeros((len(q2), 3, 2048), dtype=np.int64)
        return len(q2)

    def _walk(self, levels, L):
        """(prefix, need) per query after `levels` summed levels -- find_threshold_bin of select.hip: the bin b with
        count(bins > b) < need <= count(bins >= b)."""
        B = len(self._keys)
        prefix, need = np.zeros(B, dtype=np.uint64), np.full(B, int(L), dtype=np.int64)
        for lv in range(levels):
            nb = 1024 if lv == 2 else 2048
            for b in range(B):
                h = self._hist[b, lv, :nb]
                above = np.concatenate(([0], np.cumsum(h[::-1])))[:-1][::-1]  # elements in bins > bin
                ok = np.nonzero((above < need[b]) & (need[b] <= above + h))[0]
                binb = int(ok[0]) if len(ok) else 0
                prefix[b] = (prefix[b] << np.uint64(10 if lv == 2 else 11)) | np.uint64(binb)
                need[b] -= int(above[binb]) if len(ok) else int(h.sum() - h[0])
        return prefix, need

    def rank_cut_level(self, level, rank_limit):
        prefix, _ = self._walk(level, rank_limit)
        out = np.zeros((len(self._keys), 2048), dtype=np.int32)
        for b, keys in enumerate(self._keys):
            if level == 0:
                sel, bins = np.ones(len(keys), bool), keys >> np.uint64(21)
            elif level == 1:
                sel, bins = (keys >> np.uint64(21)) == prefix[b], (keys >> np.uint64(10)) & np.uint64(2047)
            else:
                sel, bins = (keys >> np.uint64(10)) == prefix[b], keys & np.uint64(1023)
            out[b] = np.bincount(bins[sel].astype(np.int64), minlength=2048)
        return out

    def rank_cut_level_done(self, level, hist_sum):
        self._hist[:, level, :] = np.asarray(hist_sum, dtype=np.int64)

    def rank_cut_ties(self, rank_limit):
        T, _ = self._walk(3, rank_limit)
        return np.array([(self._keys[b] == T[b]).sum() for b in range(len(self._keys))], dtype=np.int32)

    def rank_cut_finish(self, rank_limit, ties_before, k, chunk_filter=None):
        T, need_eq = self._walk(3, rank_limit)
        B, n = self._keys.shape
        r2c = np.repeat(np.arange(len(self.off) - 1), np.diff(self.off))
        S = np.full((B, k), -np.inf, np.float32)
        R = np.full((B, k), -1, np.int32)
        for b in range(B):
            eq = self._keys[b] == T[b]
            taken = eq & (np.cumsum(eq) - 1 + int(ties_before[b]) < need_eq[b])  # ties in (global) row order
            keep = (self._keys[b] > T[b]) | taken
            if chunk_filter is not None:
                keep &= np.asarray(chunk_filter, dtype=bool)[r2c]
            s, r = oracle.topk_desc(np.where(keep, self._sims[b], -np.inf), k)
            dead = (r >= 0) & ~keep[np.clip(r, 0, n - 1)]
            S[b, : len(s)] = np.where(dead, -np.inf, s)
            R[b, : len(r)] = np.where(dead, -1, r)
        return S, R

    def maxsim_topk(self, Q, k, chunk_filter=None):
        s, c = (oracle.maxsim_topk(self.E, self.off, Q, k) if chunk_filter is None
                else oracle.maxsim_topk_filtered(self.E, self.off, Q, k, chunk_filter))
        S = np.full(k, -np.inf, np.float32); C_ = np.full(k, -1, np.int32)
        S[: len(s)] = s; C_[: len(c)] = c
        return S, C_


    # -- the MaxSim batch with one candidate threshold for all shards (rl_maxsim_batch_begin / _finish), restated in NumPy: the
    # "approximate" score of a chunk is its exact one rounded down to a multiple of 4 (error < m = 4 on integer data)
    M_BOUND = 4.0

    def _all_scores(self, Qb):
        return np.stack([oracle.maxsim_scores(self.E, self.off, Q, np.float32) for Q in Qb]).astype(np.float32)

    def maxsim_topk_batch(self, Qb, k):
        outs = [self.maxsim_topk(Q, k) for Q in Qb]
        return np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs])

    def maxsim_batch_begin(self, Qb, k):
        self._exact = self._all_scores(Qb)
        self._approx = (np.floor(self._exact / 4.0) * 4.0).astype(np.float32)
        out = np.full((len(Qb), k + 1), -np.inf, np.float32)
        for b in range(len(Qb)):
            top = np.sort(self._approx[b])[::-1][:k]
            out[b, : len(top)] = top
        out[:, k] = self.M_BOUND
        self.begin_calls = getattr(self, "begin_calls", 0) + 1
        return out

    def maxsim_batch_finish(self, Qb, all_approx, rank, k):
        world, B, _ = all_approx.shape
        S = np.full((B, k), -np.inf, np.float32)
        C_ = np.full((B, k), -1, np.int32)
        self.finish_candidates = 0
        for b in range(B):
            pool = np.sort(all_approx[:, b, :k].reshape(-1))[::-1]
            A = pool[k - 1]
            thr = A - all_approx[:, b, k].max() - all_approx[rank, b, k]
            cand = np.nonzero(self._approx[b] >= thr)[0]
            self.finish_candidates += len(cand)
            order = np.lexsort((cand, -self._exact[b][cand].astype(np.float64)))[:k]
            S[b, : len(order)] = self._exact[b][cand][order]
            C_[b, : len(order)] = cand[order]
        return S, C_


def _corpus():
    rng = np.random.default_rng(42)
    off = ragged_offsets(rng, 400, 1, 9)
    E = oracle.synth_matrix(5, 400, 32, "small_int")  # integer data: heavy ties across shards
    Q = oracle.synth_matrix(6, 3, 32, "small_int")
    return E, off, Q


def _chunk_mask(n_chunks):
    return np.random.default_rng(9).random(n_chunks) < 0.3


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        E, off, Q = _corpus()
        c_lo, c_hi = shard_bounds_by_chunk(off, world)[rank]
        r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
        local_off = off[c_lo : c_hi + 1] - off[c_lo]
        local = _OracleLocal(E[r_lo:r_hi], local_off, "dot")
        sh = ShardedIndex(local, row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=local_off)
        s_rows, i_rows = sh.search_rows(Q, 25)
        s_ms, c_ms = sh.maxsim_topk(Q, 10)
        s_ch, c_ch, n_ch = sh.search_chunks(Q, 40, 6)
        ok = _chunk_mask(len(off) - 1)  # a metadata filter, evaluated on the host over the GLOBAL chunk ordinals
        f_rows = sh.search_rows(Q, 25, chunk_filter=ok)
        f_ms = sh.maxsim_topk(Q, 10, chunk_filter=ok)
        f_ch = sh.search_chunks(Q, 40, 6, chunk_filter=ok)
        b_ms = sh.maxsim_topk_batch(np.stack([Q, Q[::-1].copy()]), 10)  # a batch of two queries: one exchange for both
        # a batch of five: the shards agree on ONE candidate threshold first (an all-gather of their k best approximate scores)
        Q5 = np.stack([np.roll(Q, i, axis=0) * (1 + i % 2) for i in range(5)]).astype(np.float32)
        g_ms = sh.maxsim_topk_batch(Q5, 10)
        assert local.begin_calls == 1 and 0 < local.finish_candidates < 5 * (len(local_off) - 1) // 2  # (staged, and selective)
        # ... and when ONE shard cannot take part (an index without the image of the hi halves: RL_ERR_UNSUPPORTED from _begin) it hands
        # in an empty list and answers with its exact local top-k; the exchange stays a collective, the result the single index's
        if rank == 1:
            from raglite_amd._abi import UnsupportedError

            def refuse(Qb, k):
                raise UnsupportedError("rl_maxsim_batch_begin: this index keeps no image for the approximate pass")

            local.maxsim_batch_begin = refuse
        h_ms = sh.maxsim_topk_batch(Q5, 10)
        assert local.begin_calls == (2 if rank == 0 else 1)
        # the order-first branch: a GLOBAL cut to the 150 nearest of the 400 rows (integer data: dozens of ties ON the threshold,
        # split between the shards), then the filter, then top-25 / the two-stage search
        k_rows = sh.search_rows(Q, 25, chunk_filter=ok, rank_limit=150)
        k_ch = sh.search_chunks(Q, 40, 6, chunk_filter=ok, rank_limit=150)
        k_all = sh.search_rows(Q, 25, chunk_filter=ok, rank_limit=4000)  # a limit above the corpus: no cut
        out_q.put((rank, s_rows, i_rows, s_ms, c_ms, s_ch, c_ch, n_ch, f_rows, f_ms, f_ch, b_ms, k_rows, k_ch, k_all, g_ms, h_ms))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
def test_two_rank_gloo_matches_single_shard():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    E, off, Q = _corpus()
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    ok = _chunk_mask(len(off) - 1)
    for rank, s_rows, i_rows, s_ms, c_ms, s_ch, c_ch, n_ch, f_rows, f_ms, f_ch, b_ms, k_rows, k_ch, k_all, g_ms, h_ms in results:
        assert np.array_equal(h_ms[1], g_ms[1]) and np.array_equal(h_ms[0], g_ms[0])  # (one shard without the staged calls: same answer)
        for j in range(5):  # one threshold for all shards == the single index
            Qj = (np.roll(Q, j, axis=0) * (1 + j % 2)).astype(np.float32)
            ms, mc = oracle.maxsim_topk(E, off, Qj, 10)
            assert np.array_equal(g_ms[1][j], mc), f"rank {rank} batch query {j}: {g_ms[1][j]} vs {mc}"
            np.testing.assert_array_equal(g_ms[0][j], ms.astype(np.float32))
        for b in range(len(Q)):  # the rank cut across shards == the single-table cut (`_search.py:120-141`), ties included
            es, ei = oracle.search_rows_ranked(E, r2c, Q[b], 25, ok, 150, None, "dot", np.float32)
            assert np.array_equal(k_rows[1][b], ei), f"rank {rank} query {b} (rank cut): {k_rows[1][b][:8]} vs {ei[:8]}"
            np.testing.assert_array_equal(k_rows[0][b], es.astype(np.float32))
            cs, cc = oracle.search_chunks_ranked(E, r2c, Q[b], 40, 6, ok, 150, None, "dot", np.float32)
            assert k_ch[2][b] == len(cc) and k_ch[1][b, : len(cc)].tolist() == cc.tolist()
            np.testing.assert_array_equal(k_ch[0][b, : len(cc)], cs.astype(np.float32))
            assert np.array_equal(k_all[1][b], f_rows[1][b]) and np.array_equal(k_all[0][b], f_rows[0][b])
        for j, Qj in enumerate((Q, Q[::-1].copy())):
            ms, mc = oracle.maxsim_topk(E, off, Qj, 10)
            assert np.array_equal(b_ms[1][j], mc)
            np.testing.assert_array_equal(b_ms[0][j], ms.astype(np.float32))
        for b in range(len(Q)):  # the filtered branches across shards == the single-shard filtered oracle
            es, ei = oracle.search_rows_filtered(E, r2c, Q[b], 25, ok, "dot")
            assert np.array_equal(f_rows[1][b][: len(ei)], ei), f"rank {rank} query {b} (filtered)"
            np.testing.assert_array_equal(f_rows[0][b][: len(es)], es.astype(np.float32))
            cs, cc = oracle.search_chunks_filtered(E, r2c, Q[b], 40, 6, ok, "dot")
            assert f_ch[2][b] == len(cc) and f_ch[1][b, : len(cc)].tolist() == cc.tolist()
            np.testing.assert_array_equal(f_ch[0][b, : len(cc)], cs.astype(np.float32))
        ms, mc = oracle.maxsim_topk_filtered(E, off, Q, 10, ok)
        assert np.array_equal(f_ms[1][: len(mc)], mc)
        np.testing.assert_array_equal(f_ms[0][: len(ms)], ms.astype(np.float32))
        for b in range(len(Q)):
            es, ei = oracle.search_rows(E, Q[b], 25, "dot")
            assert np.array_equal(i_rows[b], ei), f"rank {rank} query {b}"
            np.testing.assert_array_equal(s_rows[b], es.astype(np.float32))
            cs, cc = oracle.search_chunks(E, r2c, Q[b], 40, 6, "dot")
            assert n_ch[b] == len(cc) and c_ch[b, : n_ch[b]].tolist() == cc.tolist()
            np.testing.assert_array_equal(s_ch[b, : n_ch[b]], cs.astype(np.float32))
        ms, mc = oracle.maxsim_topk(E, off, Q, 10)
        assert np.array_equal(c_ms, mc)
        np.testing.assert_array_equal(s_ms, ms.astype(np.float32))


def test_single_process_no_group():
    """world_size 1 without an initialised process group: the exchange degenerates to identity."""
    E, off, Q = _corpus()
    sh = ShardedIndex(_OracleLocal(E, off, "dot"), row_base=0, chunk_base=0, local_chunk_offsets=off)
    s, i = sh.search_rows(Q[0], 7)
    es, ei = oracle.search_rows(E, Q[0], 7, "dot")
    assert np.array_equal(i, ei)
