"""The exchange step of SURVEY.md section 8e through the C ABI: an RCCL communicator (one process per GPU) and the
all-gather + merge of every rank's local top-k, with no torch.distributed on the data path.

    id_bytes = Communicator.unique_id()            # rank 0; hand the 128 bytes to every rank (any side channel)
    comm = Communicator(rank, world, id_bytes)     # every rank (collective)
    scores, ids = comm.allgather_merge_topk(local_scores, local_ids, id_offset, k)   # CUDA tensors in and out

`Communicator.from_torch_distributed()` uses an initialised torch.distributed group only as that side channel (one
broadcast of the id at start-up); the collectives of a search step then run through librccl directly.
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from raglite_amd import _ops
from raglite_amd._abi import COMM_ID_BYTES, check, lib


class Communicator:
    def __init__(self, rank: int, world: int, unique_id: bytes) -> None:
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        _ops._ensure_init(_ops._current_device())
        handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        check(lib().rl_comm_init(C.byref(handle), int(rank), int(world), buf))
        self._handle: C.c_void_p | None = handle
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        check(lib().rl_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch_distributed(cls, group: Any = None) -> "Communicator":
        """An RCCL communicator over the ranks of an initialised torch.distributed group; torch only broadcasts the id."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box: list[Any] = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as exc:  # noqa: BLE001 - still broadcast, or the other ranks wait for ever
                box[0] = exc
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not create an RCCL unique id: {box[0]}")
        return cls(rank, world, bytes(box[0]))

    @classmethod
    def agreed(cls, group: Any = None, *, _probe=None):
        """(Communicator, None) on EVERY rank of the group, or (None, this rank's error or None) on every rank: the ranks agree -- with
        one all-reduce over torch.distributed -- that each of them can take part BEFORE the collective `rl_comm_init` is entered (a rank
        that cannot load librccl would otherwise leave the others waiting inside it), and once more that it succeeded everywhere.  What
        `bench.py --gpus N` and any multi-rank caller use, so that all ranks take the same exchange path."""
        import torch
        import torch.distributed as dist

        def all_ok(flag: bool) -> bool:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        err = None
        try:  # rank-local precondition: librccl loads and hands out an id (no collective involved)
            (_probe or cls.unique_id)()
        except Exception as exc:  # noqa: BLE001
            err = exc
        if not all_ok(err is None):
            return None, err
        comm = None
        try:
            comm = cls.from_torch_distributed(group)
        except Exception as exc:  # noqa: BLE001
            err = exc
        if not all_ok(comm is not None):
            if comm is not None:
                comm.close()
            return None, err
        return comm, None

    def close(self) -> None:
        if self._handle is not None:
            check(lib().rl_comm_destroy(self._handle))
            self._handle = None

    def __del__(self) -> None:  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter teardown
            pass

    def _args(self, scores, ids):
        a = _ops._Args()
        p_s = a.inp(scores, np.float32)
        p_i = a.inp(ids, np.int32)
        if a.mem != _ops.MEM_DEVICE:
            raise ValueError("the RCCL exchange moves device memory: pass CUDA tensors")
        s = a.keep[0]
        if s.dim() != 2 or tuple(a.keep[1].shape) != tuple(s.shape):
            raise ValueError("scores and ids must both be (n_queries, k)")
        a.ensure_device()
        return a, p_s, p_i, int(s.shape[0]), int(s.shape[1])

    def allgather_topk(self, scores, ids, id_offset: int):
        """(n_queries, k) local lists -> (world, n_queries, k) of every rank's, ids made global by `id_offset`."""
        a, p_s, p_i, nq, k = self._args(scores, ids)
        o_s, q_s = a.out((self.world, nq, k), np.float32)
        o_i, q_i = a.out((self.world, nq, k), np.int32)
        check(lib().rl_allgather_topk(self._handle, p_s, p_i, nq, k, int(id_offset), q_s, q_i, a.stream))
        return o_s, o_i

    def allgather_merge_topk(self, scores, ids, id_offset: int, k: int):
        """(n_queries, k_in) local lists -> the global top-k (n_queries, k), identical on every rank."""
        a, p_s, p_i, nq, k_in = self._args(scores, ids)
        o_s, q_s = a.out((nq, k), np.float32)
        o_i, q_i = a.out((nq, k), np.int32)
        check(lib().rl_allgather_merge_topk(self._handle, p_s, p_i, nq, k_in, int(id_offset), int(k), q_s, q_i, a.stream))
        return o_s, o_i

    def allreduce_sum_(self, t):
        """In place: t (CUDA int32 tensor) <- the sum over the ranks (`rl_allreduce_sum_u32`; the counters are non-negative)."""
        import torch

        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise ValueError("allreduce_sum_ needs a contiguous CUDA int32 tensor")
        _ops._ensure_init(t.device.index or 0)
        check(lib().rl_allreduce_sum_u32(self._handle, t.data_ptr(), t.numel(), torch.cuda.current_stream(t.device).cuda_stream))
        return t

    def allgather(self, t):
        """(world, *t.shape) of every rank's CUDA int32 tensor t (`rl_allgather_u32`)."""
        import torch

        if not (t.is_cuda and t.dtype == torch.int32):
            raise ValueError("allgather needs a CUDA int32 tensor")
        t = t.contiguous()
        out = torch.empty((self.world, *t.shape), dtype=torch.int32, device=t.device)
        _ops._ensure_init(t.device.index or 0)
        check(lib().rl_allgather_u32(self._handle, t.data_ptr(), t.numel(), out.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream))
        return out
