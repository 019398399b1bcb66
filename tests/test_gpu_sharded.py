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
