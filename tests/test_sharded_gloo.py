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

    def search_rows(self, q, k, chunk_filter=None):
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

    def maxsim_topk(self, Q, k, chunk_filter=None):
        s, c = (oracle.maxsim_topk(self.E, self.off, Q, k) if chunk_filter is None
                else oracle.maxsim_topk_filtered(self.E, self.off, Q, k, chunk_filter))
        S = np.full(k, -np.inf, np.float32); C_ = np.full(k, -1, np.int32)
        S[: len(s)] = s; C_[: len(c)] = c
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
        out_q.put((rank, s_rows, i_rows, s_ms, c_ms, s_ch, c_ch, n_ch, f_rows, f_ms, f_ch, b_ms))
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
    for rank, s_rows, i_rows, s_ms, c_ms, s_ch, c_ch, n_ch, f_rows, f_ms, f_ch, b_ms in results:
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
