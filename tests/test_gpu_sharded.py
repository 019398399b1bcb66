"""The N > 1 exchange on the device (pack -> all-gather -> unpack -> rl_merge_topk), exercised on ONE GPU by
standing in for the collective: two half-corpus shards live on the same device and a patched
`all_gather_into_tensor` hands each "rank" the other's packed list.  The result must equal the single-index search
bit for bit (merge of per-shard top-k == global top-k)."""

import numpy as np
import pytest

import raglite_amd
from raglite_amd._sharded import ShardedIndex, shard_bounds_by_chunk

pytestmark = pytest.mark.gpu


def test_device_exchange_merge_two_shards(monkeypatch):
    import torch
    import torch.distributed as dist

    raglite_amd.set_device(0)
    n, d, k, nq, qb = 40_000, 1024, 100, 32, 3
    rng = np.random.default_rng(5)
    sizes = rng.integers(1, 16, size=n)
    off = np.concatenate([[0], np.cumsum(sizes)])
    off = off[off <= n]
    if off[-1] != n:
        off = np.append(off, n)
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=11)
    Q = torch.empty((qb, nq, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=12)
    full = raglite_amd.DeviceIndex(E, off, metric="dot")
    ref_s, ref_c = full.maxsim_topk_batch(Q, k)

    bounds = shard_bounds_by_chunk(off, 2)
    shards = []
    for c_lo, c_hi in bounds:
        r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
        loc = off[c_lo : c_hi + 1] - off[c_lo]
        idx = raglite_amd.DeviceIndex(E[r_lo:r_hi], loc, metric="dot")
        shards.append(ShardedIndex(idx, row_base=r_lo, chunk_base=c_lo, local_chunk_offsets=loc))

    # what each rank would contribute to the all-gather
    packed = []
    for sh in shards:
        s, c = sh.local.maxsim_topk_batch(Q, k)
        gid = torch.where(c >= 0, c + sh.chunk_base, torch.full_like(c, -1)).to(torch.int32)
        packed.append(torch.stack([s.contiguous().view(torch.int32), gid], dim=-1).contiguous())

    def fake_all_gather(out, inp, group=None):
        if inp.dim() == 2 and inp.shape[1] == k + 1:
            # the threshold exchange of a batch of three or more (rl_maxsim_batch_begin): these shards are too small to keep an image of
            # the hi halves, so each contributes an empty list (-inf, bound 0) and answers with its exact local top-k
            assert bool((inp[:, :k].view(torch.float32) == float("-inf")).all()) and bool((inp[:, k] == 0).all())
            stacked = out.view(2, *inp.shape)
            stacked[0].copy_(inp)
            stacked[1].copy_(inp)
            return
        # the calling rank's own slot must carry what it passed in; the peer's slot the peer's list
        me = 0 if torch.equal(inp, packed[0]) else 1
        assert torch.equal(inp, packed[me])
        stacked = out.view(2, *inp.shape)  # the library allocates the concatenated form every backend accepts
        stacked[0].copy_(packed[0])
        stacked[1].copy_(packed[1])

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    for sh in shards:
        s, c = sh.maxsim_topk_batch(Q, k)
        assert s.is_cuda and c.is_cuda  # stayed on the device
        assert torch.equal(s, ref_s) and torch.equal(c.to(ref_c.dtype), ref_c)
    monkeypatch.undo()
    # the batched row search (cfg 5's 8-GPU shape) takes the same device exchange
    Qr = torch.empty((130, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Qr, seed=13)
    full_c = raglite_amd.DeviceIndex(E, off, metric="cosine")
    ref_rs, ref_rr = full_c.search_rows(Qr, k)
    row_shards, packed_r = [], []
    for c_lo, c_hi in bounds:
        r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
        idx = raglite_amd.DeviceIndex(E[r_lo:r_hi], metric="cosine")
        row_shards.append(ShardedIndex(idx, row_base=r_lo, chunk_base=c_lo))
        s, r = idx.search_rows(Qr, k)
        gid = torch.where(r >= 0, r + r_lo, torch.full_like(r, -1)).to(torch.int32)
        packed_r.append(torch.stack([s.contiguous().view(torch.int32), gid], dim=-1).contiguous())

    def fake_rows(out, inp, group=None):
        stacked = out.view(2, *inp.shape)
        stacked[0].copy_(packed_r[0])
        stacked[1].copy_(packed_r[1])

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_rows)
    for sh in row_shards:
        s, r = sh.search_rows(Qr, k)
        assert s.is_cuda and torch.equal(s, ref_rs) and torch.equal(r.to(ref_rr.dtype), ref_rr)
    monkeypatch.undo()
    for sh in row_shards:
        sh.local.close()
    full_c.close()
    # world = 1: global ids only, still on the device
    one = ShardedIndex(full, row_base=0, chunk_base=7, local_chunk_offsets=off)
    s, c = one.maxsim_topk_batch(Q, k)
    assert s.is_cuda and torch.equal(s, ref_s) and torch.equal(c.to(ref_c.dtype), ref_c + 7)
    for sh in shards:
        sh.local.close()
    full.close()


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path end to end (what the driver launches, minus RCCL): two ranks share cuda:0, the all-gather
    runs over gloo, rank 0 prints ONE JSON line for n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    import socket

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "100000",
           "--same-gpu", "--backend", "gloo"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["roofline"]["algorithmic_bytes_per_launch"] < 4.0 * 100000 * 1024  # a shard, not the whole corpus


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (how the driver's scaling run may call it): bench.py re-executes itself as two
    ranks through torch.distributed.run; n_gpus in the JSON line comes from the process group."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "200000", "--same-gpu",
           "--backend", "gloo"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    # (a launch covers all passes of a step -- grid row = pass -- over this rank's shard, not over the whole corpus)
    assert out["roofline"]["algorithmic_bytes_per_launch"] / out["roofline"].get("passes_per_launch", 1) < 4.0 * 200000 * 1024


def test_bench_gpus_n_refuses_a_box_with_fewer_gpus():
    """`--gpus 8` on a one-GPU box is an error (non-zero exit), never a silent single-GPU run that prints n_gpus: 1."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    import torch

    root = Path(__file__).resolve().parent.parent
    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "visible" in res.stderr and not any(ln.startswith("{") for ln in res.stdout.splitlines())


def test_rccl_communicator_world_of_one():
    """The exchange step behind the C ABI (rl_comm_* / rl_allgather_topk / rl_allgather_merge_topk over librccl) on the one
    GPU a test box has: communicator init, the all-gather and the merge with a single rank.  (Two ranks cannot share a
    device under RCCL; the two-rank logic is covered over gloo in tests/test_sharded_gloo.py.)"""
    import torch

    raglite_amd.set_device(0)
    comm = raglite_amd.Communicator(0, 1, raglite_amd.Communicator.unique_id())
    assert (comm.rank, comm.world) == (0, 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    scores = torch.randn((7, 50), device="cuda", generator=g).sort(dim=1, descending=True).values
    ids = torch.stack([torch.randperm(1000, device="cuda", generator=g)[:50] for _ in range(7)]).to(torch.int32)
    ids[3, 40:] = -1  # padding of a short list
    scores[3, 40:] = float("-inf")
    gs, gi = comm.allgather_topk(scores, ids, 5000)
    assert gs.shape == (1, 7, 50) and torch.equal(gs[0], scores)
    assert torch.equal(gi[0], torch.where(ids >= 0, ids + 5000, torch.full_like(ids, -1)))
    ms, mi = comm.allgather_merge_topk(scores, ids, 5000, 20)
    # one list per query: the merge is its (score desc, id asc) top-20
    want_s, want_i = raglite_amd.merge_topk(gs, gi, 20)
    assert torch.equal(ms, want_s) and torch.equal(mi, want_i)
    assert torch.equal(ms, scores[:, :20]) or bool((scores[:, :-1] == scores[:, 1:]).any())
    # the sharded index on top of it: global ids, results stay on the device
    n, d = 30_000, 256
    rng = np.random.default_rng(8)
    off = np.concatenate(([0], np.cumsum(rng.integers(1, 10, size=n))))
    off = np.concatenate((off[off < n], [n])).astype(np.int64)
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=21, kind="small_int")
    Q = torch.empty((5, 32, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=22, kind="small_int")
    idx = raglite_amd.DeviceIndex(E, off, metric="dot")
    sh = ShardedIndex(idx, row_base=1000, chunk_base=77, local_chunk_offsets=off, comm=comm)
    ref_s, ref_c = idx.maxsim_topk_batch(Q, 64)
    s, c = sh.maxsim_topk_batch(Q, 64)
    assert s.is_cuda and torch.equal(s, ref_s) and torch.equal(c, ref_c + 77)
    qs = Q[:, 0, :].contiguous()
    rs, rr = idx.search_rows(qs, 30)
    s2, r2 = sh.search_rows(qs, 30)
    assert torch.equal(s2, rs) and torch.equal(r2, rr + 1000)
    cs, cc, cn = idx.search_chunks(qs, 40, 10)
    s3, c3, n3 = sh.search_chunks(qs, 40, 10)  # CUDA queries through the two-stage search
    assert s3.is_cuda and torch.equal(n3, cn) and torch.equal(s3, cs)
    assert torch.equal(c3, torch.where(cc >= 0, cc + 77, torch.full_like(cc, -1)))
    s4, c4, n4 = sh.search_chunks(qs[0], 40, 10)
    assert torch.equal(s4, cs[0]) and int(n4) == int(cn[0])
    idx.close()
    comm.close()


def test_device_search_chunks_two_shards(monkeypatch):
    """`ShardedIndex.search_chunks` with CUDA queries across two shards on one device (the collective stood in for as
    above): the merged top-`num_hits` rows grouped by chunk equal the single-index two-stage search, bit for bit."""
    import torch
    import torch.distributed as dist

    raglite_amd.set_device(0)
    n, d, num_hits, k = 20_000, 128, 60, 12
    rng = np.random.default_rng(15)
    off = np.concatenate(([0], np.cumsum(rng.integers(1, 8, size=n))))
    off = np.concatenate((off[off < n], [n])).astype(np.int64)
    E = torch.empty((n, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(E, seed=31, kind="small_int")  # integer data: heavy ties across the shard boundary
    Q = torch.empty((9, d), dtype=torch.float32, device="cuda")
    raglite_amd.synth_fill(Q, seed=32, kind="small_int")
    full = raglite_amd.DeviceIndex(E, off, metric="dot")
    ref_s, ref_c, ref_n = full.search_chunks(Q, num_hits, k)
    shards, contrib = [], {}
    for c_lo, c_hi in shard_bounds_by_chunk(off, 2):
        r_lo, r_hi = int(off[c_lo]), int(off[c_hi])
        loc = off[c_lo : c_hi + 1] - off[c_lo]
        shards.append(ShardedIndex(raglite_amd.DeviceIndex(E[r_lo:r_hi], loc, metric="dot"), row_base=r_lo, chunk_base=c_lo,
                                   local_chunk_offsets=loc))
    # every rank's contributions to the two gathers (rows, chunks), keyed by what the rank passes in
    for kind in ("rows", "chunks"):
        packs = []
        for sh in shards:
            s, r = sh.local.search_rows(Q, num_hits)
            if kind == "rows":
                ids = torch.where(r >= 0, r + sh.row_base, torch.full_like(r, -1))
            else:
                loc = torch.as_tensor(sh.local_chunk_offsets, device="cuda")
                cl = torch.searchsorted(loc, r.to(torch.int64).clamp(min=0), right=True) - 1
                ids = torch.where(r >= 0, cl + sh.chunk_base, torch.full_like(cl, -1))
            packs.append(torch.stack([s.contiguous().view(torch.int32), ids.to(torch.int32)], dim=-1).contiguous())
        contrib[kind] = packs

    def fake_all_gather(out, inp, group=None):
        for packs in contrib.values():
            if any(torch.equal(inp, p) for p in packs):
                stacked = out.view(2, *inp.shape)
                stacked[0].copy_(packs[0])
                stacked[1].copy_(packs[1])
                return
        raise AssertionError("unexpected all-gather input")

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather)
    for sh in shards:
        s, c, cnt = sh.search_chunks(Q, num_hits, k)
        assert s.is_cuda and torch.equal(cnt, ref_n) and torch.equal(s, ref_s) and torch.equal(c, ref_c)
    monkeypatch.undo()
    for sh in shards:
        sh.local.close()
    full.close()
