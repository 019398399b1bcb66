"""Multi-GPU readiness without the hardware: EIGHT processes over gloo run `ShardedIndex` with uneven shards and one EMPTY shard through
everything north_star's 8-GPU split exercises -- the MaxSim batch with its threshold exchange, the two-stage search behind a GLOBAL
rank cut, the merge of more candidates than the merge kernel sorts (world x k > 8192: `merge_order_torch`) -- and through the two
failure protocols: a rank that cannot build its RCCL communicator makes EVERY rank take the torch.distributed path
(`Communicator.agreed`), and a rank whose local step raises still enters every collective of the call, so nobody hangs.

The local searcher is the oracle-backed double of tests/test_sharded_gloo.py; the merged results must equal the single-index oracle
(`/root/reference/src/raglite/_search.py:66-79,120-149` over one table) bit for bit."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle
from raglite_amd._comm import Communicator
from raglite_amd._sharded import ShardedIndex, _merge_order, merge_order_torch
from tests.test_sharded_gloo import _OracleLocal, _free_port
from tests.util import ragged_offsets

WORLD = 8
# chunk cut points of the 8 shards: uneven (6 to 56 chunks), shard 3 EMPTY
CUTS = [0, 11, 67, 80, 80, 101, 107, 139, 150]


class _Local(_OracleLocal):
    """The double, tolerant of a shard without rows."""

    def search_rows(self, q, k, chunk_filter=None, rank_limit=None):
        if len(self.E) == 0:
            q2 = np.atleast_2d(q)
            S, I = np.full((len(q2), k), -np.inf, np.float32), np.full((len(q2), k), -1, np.int32)
            return (S[0], I[0]) if np.ndim(q) == 1 else (S, I)
        return super().search_rows(q, k, chunk_filter, rank_limit)

    def maxsim_topk(self, Q, k, chunk_filter=None):
        if len(self.E) == 0:
            return np.full(k, -np.inf, np.float32), np.full(k, -1, np.int32)
        return super().maxsim_topk(Q, k, chunk_filter)

    def _all_scores(self, Qb):
        if len(self.E) == 0:
            return np.zeros((len(Qb), 0), np.float32)
        return super()._all_scores(Qb)

    def rank_cut_begin(self, queries):
        if len(self.E) == 0:
            q2 = np.atleast_2d(queries)
            self._sims = np.zeros((len(q2), 0), np.float32)
            self._keys = np.zeros((len(q2), 0), np.uint64)
            self._hist = np.zeros((len(q2), 3, 2048), dtype=np.int64)
            return len(q2)
        return super().rank_cut_begin(queries)

    def rank_cut_finish(self, rank_limit, ties_before, k, chunk_filter=None):
        if len(self.E) == 0:
            B = len(self._keys)
            return np.full((B, k), -np.inf, np.float32), np.full((B, k), -1, np.int32)
        return super().rank_cut_finish(rank_limit, ties_before, k, chunk_filter)


def _corpus():
    rng = np.random.default_rng(4242)
    off = ragged_offsets(rng, 1200, 1, 9)[: CUTS[-1] + 1]
    assert len(off) == CUTS[-1] + 1
    n = int(off[-1])
    E = oracle.synth_matrix(15, n, 32, "small_int")  # integer data: heavy ties across shards
    Q = oracle.synth_matrix(16, 3, 32, "small_int")
    Q5 = np.stack([np.roll(Q, i, axis=0) * (1 + i % 2) for i in range(5)]).astype(np.float32)
    return E, off, Q, Q5


def _mask(n_chunks):
    return np.random.default_rng(19).random(n_chunks) < 0.4


def _worker(rank, port, out_q):
    try:
        _worker_body(rank, port, out_q)
    except BaseException as exc:  # noqa: BLE001 - report instead of leaving the parent waiting for its queue
        import traceback

        out_q.put({"rank": rank, "error": "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))})
        raise


def _worker_body(rank, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        E, off, Q, Q5 = _corpus()
        c_lo, c_hi = CUTS[rank], CUTS[rank + 1]
        r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
        local_off = off[c_lo : c_hi + 1] - off[c_lo]
        local = _Local(E[r_lo:r_hi], local_off, "dot")
        sh = ShardedIndex(local, row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=local_off)
        out = {"rank": rank}
        # (1) the MaxSim batch: one candidate threshold for all shards (the empty shard hands in an empty list)
        out["batch"] = sh.maxsim_topk_batch(Q5, 10)
        out["staged"] = getattr(local, "begin_calls", 0)
        # (2) two-stage search behind a GLOBAL rank cut + filter, and the row search behind it
        ok = _mask(len(off) - 1)
        out["cut_rows"] = sh.search_rows(Q, 25, chunk_filter=ok, rank_limit=150)
        out["cut_chunks"] = sh.search_chunks(Q, 40, 6, chunk_filter=ok, rank_limit=150)
        # ... and with a score-matrix budget that holds ONE query of the largest shard: every rank splits the batch the same way
        # (three sub-batches, three rounds of collectives each), the empty shard included
        sh.rank_cut_scratch_bytes = 4 * 420 + 64
        out["cut_rows_split"] = sh.search_rows(Q, 25, chunk_filter=ok, rank_limit=150)
        sh.rank_cut_scratch_bytes = ShardedIndex.rank_cut_scratch_bytes
        # (3) more candidates than rl_merge_topk's kernel sorts: WORLD x 1100 = 8800 > 8192 takes merge_order_torch (torch tensors over
        # gloo stand in for the device tensors; the code path is the device one)
        s_np, i_np = local.search_rows(Q, 1100)
        ms, mi = sh._exchange_merge_device(torch.from_numpy(s_np.copy()), torch.from_numpy(i_np.astype(np.int64)), r_lo, 1100)  # noqa: SLF001
        out["big_merge"] = (ms.numpy(), mi.numpy())
        # (4) one rank cannot take part in the RCCL communicator: every rank must end up WITHOUT one
        def probe():
            if rank == 5:
                raise OSError("librccl not found (injected)")
            return b""

        comm, err = Communicator.agreed(_probe=probe)
        out["comm_none"] = comm is None
        out["comm_err"] = type(err).__name__ if err is not None else None
        # (5) one rank's local step raises something that is NOT "unsupported": it still enters both collectives of the call, then raises;
        # the others finish the call and learn from the bound column that a shard is missing
        if rank == 2:
            def boom(Qb, k):
                raise MemoryError("hipMalloc failed (injected)")

            local.maxsim_batch_begin = boom
        try:
            sh.maxsim_topk_batch(Q5, 10)
            out["failure"] = "no error"
        except MemoryError as exc:
            out["failure"] = f"MemoryError: {exc}"
        except RuntimeError as exc:
            out["failure"] = f"RuntimeError: {exc}"
        # (6) the same failure on the route WITHOUT a threshold exchange (fp16 queries over fp16-stored shards: the local batch goes
        # straight to the merge all-gather) -- round 5 let the exception escape before the collective and the other ranks hung in it.
        # Torch tensors over gloo stand in for the device tensors (the code path is the device one: the merge comes back poisoned).
        local.storage = "f16"
        if rank == 2:
            local.maxsim_batch_begin = _Local.maxsim_batch_begin.__get__(local)

            def boom16(Qb, k):
                raise MemoryError("first-call image build failed (injected)")

            local.maxsim_topk_batch = boom16
        try:
            sh.maxsim_topk_batch(Q5.astype(np.float16), 10)
            out["failure16"] = "no error"
        except MemoryError as exc:
            out["failure16"] = f"MemoryError: {exc}"
        except RuntimeError as exc:
            out["failure16"] = f"RuntimeError: {exc}"
        out["staged_after_16"] = getattr(local, "begin_calls", 0)
        # (the merge itself, 8800 records wide so that CPU tensors can take it: rank 2 sends the marker a failed rank sends)
        if rank != 2:
            ms, mi = sh._exchange_merge_device(torch.from_numpy(s_np.copy()), torch.from_numpy(i_np.astype(np.int32)), r_lo, 1100)  # noqa: SLF001
        else:
            ms, mi = sh._exchange_merge_device(torch.full((len(Q), 1100), float("-inf")), torch.full((len(Q), 1100), -2, dtype=torch.int32), r_lo, 1100)  # noqa: SLF001
        out["poisoned"] = bool(torch.isnan(ms).all()) and bool((mi == -1).all())
        # ... and the group is still usable afterwards
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t)
        out["after"] = int(t.item())
        out_q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_ranks_uneven_and_empty_shards_match_the_single_index():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = []
    for _ in range(WORLD):
        results.append(q.get(timeout=240))
        assert "error" not in results[-1], results[-1]["error"]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    E, off, Q, Q5 = _corpus()
    r2c = np.repeat(np.arange(len(off) - 1), np.diff(off))
    ok = _mask(len(off) - 1)
    want_big = [oracle.search_rows(E, Q[b], 1100, "dot") for b in range(len(Q))]
    for out in sorted(results, key=lambda o: o["rank"]):
        rank = out["rank"]
        assert out["staged"] == 1  # the threshold exchange ran (once per rank, the empty shard included)
        for j in range(5):
            ms, mc = oracle.maxsim_topk(E, off, Q5[j], 10)
            assert np.array_equal(out["batch"][1][j], mc), f"rank {rank} batch query {j}"
            np.testing.assert_array_equal(out["batch"][0][j], ms.astype(np.float32))
        for b in range(len(Q)):
            es, ei = oracle.search_rows_ranked(E, r2c, Q[b], 25, ok, 150, None, "dot", np.float32)
            assert np.array_equal(out["cut_rows"][1][b], ei), f"rank {rank} query {b} (rank cut)"
            np.testing.assert_array_equal(out["cut_rows"][0][b], es.astype(np.float32))
            assert np.array_equal(out["cut_rows_split"][1][b], ei) and np.array_equal(out["cut_rows_split"][0][b], out["cut_rows"][0][b])
            cs, cc = oracle.search_chunks_ranked(E, r2c, Q[b], 40, 6, ok, 150, None, "dot", np.float32)
            got = out["cut_chunks"]
            assert got[2][b] == len(cc) and got[1][b, : len(cc)].tolist() == cc.tolist()
            np.testing.assert_array_equal(got[0][b, : len(cc)], cs.astype(np.float32))
            ws, wi = want_big[b]
            n_valid = len(wi)  # fewer than 1100 rows exist: the tail is padding
            assert np.array_equal(out["big_merge"][1][b, :n_valid], wi)
            np.testing.assert_array_equal(out["big_merge"][0][b, :n_valid], ws.astype(np.float32))
            assert (out["big_merge"][1][b, n_valid:] == -1).all()
        assert out["comm_none"], f"rank {rank} kept a communicator the others do not have"
        assert out["comm_err"] == ("OSError" if rank == 5 else None)
        if rank == 2:
            assert out["failure"].startswith("MemoryError: hipMalloc failed")
        else:
            assert out["failure"].startswith("RuntimeError: ShardedIndex.maxsim_topk_batch: another rank failed")
        if rank == 2:
            assert out["failure16"].startswith("MemoryError: first-call image build failed")
        else:
            assert out["failure16"].startswith("RuntimeError: ShardedIndex.maxsim_topk_batch: another rank failed")
        assert out["staged_after_16"] == out["staged"] + (1 if rank != 2 else 0)  # (5) staged once more on the healthy ranks, (6) never
        assert out["poisoned"], f"rank {rank}: a merge that lacks a shard came back looking like an answer"
        assert out["after"] == sum(range(WORLD))


def test_merge_order_torch_equals_the_host_order_on_ties_padding_and_nan():
    """The big-merge path's ordering on its own: (score desc, id asc), padding (-1) and NaN last -- torch and NumPy agree."""
    rng = np.random.default_rng(3)
    B, n, k = 4, 9000, 1200
    s = rng.integers(-3, 4, size=(B, n)).astype(np.float32)
    i = np.stack([rng.permutation(n) for _ in range(B)]).astype(np.int64)
    s[0, :50] = np.nan
    i[1, 100:400] = -1
    order_t, nv_t = merge_order_torch(torch.from_numpy(s), torch.from_numpy(i), k)
    order_n, nv_n = _merge_order(s, i, k)
    assert np.array_equal(nv_t.numpy(), nv_n)
    for b in range(B):
        v = int(nv_n[b])
        assert np.array_equal(np.take_along_axis(i[b], order_t.numpy()[b], 0)[:v], np.take_along_axis(i[b], order_n[b], 0)[:v])


class _ReplayComm:
    """A Communicator double for ONE process: world = 2, no torch.distributed.  `allgather` records what this rank hands in and answers
    with both ranks' arrays for the exchanges whose partner array is already known from an earlier pass (a dummy -- twice its own array
    -- otherwise), so that a few passes over both ranks converge on exactly what two real ranks would have exchanged."""

    world = 2

    def __init__(self, rank, known):
        self.rank, self.known, self.mine, self.complete = rank, known, [], True

    def allgather(self, t):
        j = len(self.mine)
        self.mine.append(t.clone())
        other = self.known[1 - self.rank][j] if j < len(self.known[1 - self.rank]) else None
        if other is None or other.shape != t.shape:
            self.complete = False
            other = t
        parts = [t, other] if self.rank == 0 else [other, t]
        return torch.stack(parts)


def test_communicator_only_index_merges_host_arrays_over_all_ranks():
    """ShardedIndex(comm=Communicator(world > 1)) WITHOUT torch.distributed and with host (NumPy) queries: `search_rows`, `maxsim_topk` and
    the host path of `maxsim_topk_batch` must exchange through the communicator.  (Round 4's `_exchange_host` went to the torch-only
    gather, read "world of one" there and returned the local shard's lists as the global top-k.)"""
    assert not (dist.is_available() and dist.is_initialized())
    rng = np.random.default_rng(77)
    off = ragged_offsets(rng, 400, 1, 9)
    n = int(off[-1])
    E = oracle.synth_matrix(21, n, 32, "small_int")
    Q = oracle.synth_matrix(22, 3, 32, "small_int")
    Q5 = np.stack([np.roll(Q, i, axis=0) * (1 + i % 2) for i in range(5)]).astype(np.float32)
    cut = (len(off) - 1) // 3  # uneven: a third / two thirds of the chunks
    bounds = [(0, cut), (cut, len(off) - 1)]

    def run(call):
        known = [[], []]
        for _ in range(4):  # (a call makes at most two exchanges: two passes fix them, the third confirms)
            outs, comms = [], []
            for rank, (c_lo, c_hi) in enumerate(bounds):
                r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
                local_off = off[c_lo : c_hi + 1] - off[c_lo]
                comm = _ReplayComm(rank, known)
                sh = ShardedIndex(_Local(E[r_lo:r_hi], local_off, "dot"), row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=local_off, comm=comm)
                outs.append(call(sh))
                comms.append(comm)
            known = [c.mine for c in comms]
            if all(c.complete for c in comms):
                assert all(len(c.mine) >= 1 for c in comms), "no exchange went through the communicator"
                return outs
        raise AssertionError("the replayed exchanges did not converge")

    for rank_out in run(lambda sh: sh.search_rows(Q, 25)):
        for b in range(len(Q)):
            es, ei = oracle.search_rows(E, Q[b], 25, "dot", np.float32)
            assert np.array_equal(rank_out[1][b], ei)
            np.testing.assert_array_equal(rank_out[0][b], np.asarray(es, np.float32))
    for rank_out in run(lambda sh: sh.maxsim_topk(Q, 10)):
        ms, mc = oracle.maxsim_topk(E, off, Q, 10)
        assert np.array_equal(rank_out[1], mc)
        np.testing.assert_array_equal(rank_out[0], ms.astype(np.float32))
    for rank_out in run(lambda sh: sh.maxsim_topk_batch(Q5, 10)):
        for j in range(5):
            ms, mc = oracle.maxsim_topk(E, off, Q5[j], 10)
            assert np.array_equal(rank_out[1][j], mc), f"batch query {j}"
            np.testing.assert_array_equal(rank_out[0][j], ms.astype(np.float32))
    # several ranks and NO transport for host arrays is an error, never a silent world of one
    class _NoTransport:
        world, rank = 2, 0
        allgather = None

    sh = ShardedIndex(_Local(E[: int(off[cut])], off[: cut + 1], "dot"), row_base=0, chunk_base=0, local_chunk_offsets=off[: cut + 1])
    sh._world = lambda: 2  # noqa: SLF001
    with pytest.raises(RuntimeError, match="neither torch.distributed nor a Communicator"):
        sh.search_rows(Q, 5)


def test_fp16_queries_over_fp16_stored_shards_skip_the_threshold_exchange():
    """`ShardedIndex.maxsim_topk_batch`: with fp16 queries and fp16-stored shards every shard's pass is exact, so the staged form (begin ->
    all-gather of approximate lists -> finish) is not entered; any other combination still takes it."""
    calls = []

    class _Local:
        storage = "f16"

        def maxsim_topk_batch(self, Qb, k):
            calls.append(("batch", str(Qb.dtype)))
            return np.zeros((len(Qb), k), np.float32), np.zeros((len(Qb), k), np.int32)

        def maxsim_batch_begin(self, Qb, k):
            calls.append(("begin", str(Qb.dtype)))
            raise RuntimeError("staged")

    sh = ShardedIndex(_Local(), row_base=0, chunk_base=0)
    sh._world = lambda: 8  # noqa: SLF001
    sh._local_maxsim_batch(np.zeros((8, 4, 32), np.float16), 5)  # noqa: SLF001
    assert calls == [("batch", "float16")]
    calls.clear()
    sh._local_maxsim_batch(torch.zeros((8, 4, 32), dtype=torch.float16), 5)  # noqa: SLF001
    assert calls == [("batch", "torch.float16")]
    calls.clear()
    sh._allgather_int = lambda x: np.stack([x] * 8)  # noqa: SLF001
    sh._local_maxsim_batch(np.zeros((8, 4, 32), np.float32), 5)  # noqa: SLF001
    assert calls[0] == ("begin", "float32")
    calls.clear()
    sh._local_maxsim_batch(torch.zeros((8, 4, 32), dtype=torch.bfloat16), 5)  # noqa: SLF001  (bfloat16 is not IEEE fp16: staged like fp32)
    assert calls[0] == ("begin", "torch.bfloat16")
