"""ctypes binding of libraglite_hip.so (the C ABI declared in include/raglite_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Status codes map to the reference's error conventions:
RL_ERR_INVALID / RL_ERR_UNSUPPORTED -> ValueError, RL_ERR_NOMEM -> MemoryError, RL_ERR_HIP ->
RuntimeError.  ctypes releases the GIL for the duration of every call, which is what the
reference's threaded callers need (`src/raglite/_insert.py:208-237`, `src/raglite/_rag.py:317`).
"""

from __future__ import annotations

import ctypes as C
import functools
import os
from pathlib import Path

# RAGLITE_HIP_LIB: developer hook of the PYTHON loader for same-box A/B timing of kernel variants and for the experiments build
# (libraglite_hip_exp.so, scripts/gpu_calls/); unset in normal use.  The library itself reads no environment variable.
LIB_PATH = Path(os.environ.get("RAGLITE_HIP_LIB") or Path(__file__).resolve().parent / "_lib" / "libraglite_hip.so")

RL_OK, RL_ERR_INVALID, RL_ERR_HIP, RL_ERR_UNSUPPORTED, RL_ERR_NOMEM = 0, -1, -2, -3, -4
MEM_HOST, MEM_DEVICE = 0, 1
METRICS = {"cosine": 0, "dot": 1, "l2": 2}
SYNTH_KINDS = {"uniform": 0, "small_int": 1}
# rl_option (include/raglite_hip.h "options"): route switches of an index; the library reads no environment variable
OPTIONS = {
    "hi_search": 1, "hi_maxsim": 2, "hi_products": 3, "pp_pass": 4, "fused_topk": 5, "fused_hi": 6, "fused_pp": 7, "fused_topk_cap": 8,
    "fused_topk_stride": 9, "gemm_pass": 10, "query_pairs": 11, "planes_gemm": 12, "keep_image": 13, "keep_hi": 14,
    "image_headroom_mb": 15, "arithmetic": 16, "exact_kth_threshold": 17, "fused_two_rounds": 18,
    "keep_hi_plane": 19, "pairs_packed": 20, "f16_exact": 21, "lazy_images": 22, "fused_pp_sample": 23, "list_select": 24, "hi_few": 25, "topk_block": 26, "hi_pivot": 27,
}

c_void_p, c_int, c_i32, c_i64, c_u64, c_size_t = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
c_double, c_char_p = C.c_double, C.c_char_p

# name -> argtypes (every function returns int status unless listed in _RESTYPES)
_SIGNATURES = {
    "rl_version": [],
    "rl_last_error": [],
    "rl_init": [c_int],
    "rl_device_count": [C.POINTER(c_int)],
    "rl_device_info": [c_int, c_char_p, c_int, C.POINTER(c_int), C.POINTER(c_i64)],
    "rl_dev_alloc": [C.POINTER(c_void_p), c_size_t],
    "rl_dev_free": [c_void_p],
    "rl_memcpy_h2d": [c_void_p, c_void_p, c_size_t, c_void_p],
    "rl_memcpy_d2h": [c_void_p, c_void_p, c_size_t, c_void_p],
    "rl_memcpy_d2d": [c_void_p, c_void_p, c_size_t, c_void_p],
    "rl_stream_sync": [c_void_p],
    "rl_synth_fill": [c_void_p, c_i64, c_i64, c_u64, c_int, c_void_p],
    "rl_pool_norm": [c_void_p, c_i64, c_i32, c_void_p, c_void_p, c_i64, c_i32, c_double, c_void_p, c_void_p,
                     c_int, c_void_p],
    "rl_adapter_apply": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_index_create": [C.POINTER(c_void_p), c_void_p, c_i64, c_i32, c_void_p, c_i64, c_int, c_int, c_void_p],
    "rl_index_destroy": [c_void_p],
    "rl_index_info": [c_void_p, C.POINTER(c_i64), C.POINTER(c_i32), C.POINTER(c_i64), C.POINTER(c_int)],
    "rl_index_memory": [c_void_p, C.POINTER(c_i64)],
    "rl_index_prepare": [c_void_p, C.c_uint32, C.POINTER(C.c_uint32), c_void_p],
    "rl_partition_similarity": [c_void_p, c_i64, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p],
    "rl_chunk_best_rows": [c_void_p, c_void_p, c_i32, c_void_p, c_i32, c_void_p, c_int, c_void_p],
    "rl_gather_rows": [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "rl_index_create_f16": [C.POINTER(c_void_p), c_void_p, c_i64, c_i32, c_void_p, c_i64, c_int, c_int, c_void_p],
    "rl_index_append": [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p],
    "rl_index_delete_chunks": [c_void_p, c_void_p, c_i64, c_void_p],
    "rl_index_live": [c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), c_void_p],
    "rl_index_compact": [c_void_p, c_void_p, C.POINTER(c_i64), C.POINTER(c_i64), c_void_p],
    "rl_index_filter_stats": [c_void_p, C.POINTER(c_i64), c_void_p],
    "rl_index_set_arithmetic": [c_void_p, c_int],
    "rl_index_arithmetic": [c_void_p, C.POINTER(c_int)],
    "rl_set_default_option": [c_int, c_i64],
    "rl_get_default_option": [c_int, C.POINTER(c_i64)],
    "rl_index_set_option": [c_void_p, c_int, c_i64],
    "rl_index_get_option": [c_void_p, c_int, C.POINTER(c_i64)],
    "rl_search_rows_filtered": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "rl_search_chunks_filtered": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_void_p],
    "rl_maxsim_topk_filtered": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "rl_search_rows_ranked": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_int, c_void_p],
    "rl_search_chunks_ranked": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
                                c_int, c_void_p],
    "rl_search_rows": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_search_chunks": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_topk": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_scores": [c_void_p, c_void_p, c_i32, c_void_p, c_int, c_void_p],
    "rl_maxsim_topk_batch": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_topk_batch_f16": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_approx_scores": [c_void_p, c_void_p, c_i32, c_i32, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_batch_begin": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_int, c_void_p],
    "rl_maxsim_batch_finish": [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_maxsim_rerank": [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_i32, c_void_p, c_int, c_void_p],
    "rl_merge_topk": [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_topk": [c_void_p, c_i32, c_i64, c_i64, c_i32, c_void_p, c_void_p, c_int, c_void_p],
    "rl_time_kernel": [c_void_p, c_int, c_void_p, c_i32, c_i32, C.POINTER(C.c_float), c_void_p],
    "rl_comm_unique_id": [c_void_p],
    "rl_comm_init": [C.POINTER(c_void_p), c_int, c_int, c_void_p],
    "rl_comm_info": [c_void_p, C.POINTER(c_int), C.POINTER(c_int)],
    "rl_comm_destroy": [c_void_p],
    "rl_allgather_topk": [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p],
    "rl_allgather_merge_topk": [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p],
    "rl_allreduce_sum_u32": [c_void_p, c_void_p, c_i64, c_void_p],
    "rl_allgather_u32": [c_void_p, c_void_p, c_i64, c_void_p, c_void_p],
    "rl_rank_cut_begin": [c_void_p, c_void_p, c_i32, c_int, c_void_p],
    "rl_rank_cut_level": [c_void_p, c_int, c_i64, c_void_p, c_int, c_void_p],
    "rl_rank_cut_level_done": [c_void_p, c_int, c_void_p, c_int, c_void_p],
    "rl_rank_cut_ties": [c_void_p, c_i64, c_void_p, c_int, c_void_p],
    "rl_rank_cut_finish": [c_void_p, c_i64, c_void_p, c_void_p, c_i32, c_void_p, c_void_p, c_int, c_void_p],
}
COMM_ID_BYTES = 128
_RESTYPES = {"rl_last_error": c_char_p}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class RagliteHipError(RuntimeError):
    """HIP runtime failure inside libraglite_hip.so."""


class UnsupportedError(ValueError):
    """RL_ERR_UNSUPPORTED: the arguments are valid but this entry point does not cover them (a ValueError, as before; its own type so
    that callers with another path -- `ShardedIndex.maxsim_topk_batch` -- can take it)."""


@functools.lru_cache(maxsize=1)
def lib() -> C.CDLL:
    """Load the library (once).  Fails loudly: the product path has no CPU fallback."""
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m raglite_amd._build` (needs hipcc; "
            "cross-compiles for gfx950 without a GPU). raglite_amd has no CPU fallback."
        )
    try:
        # PyTorch-ROCm bundles its own libamdhip64; loading it FIRST makes this library bind to the same HIP
        # runtime (same soname), which is what lets device pointers and streams cross between the two.  With the
        # opposite order two runtimes end up in the process and the second one finds no device.
        import torch  # noqa: F401
    except ImportError:  # pure-ctypes use without PyTorch: the system ROCm runtime is loaded instead
        pass
    handle = C.CDLL(str(LIB_PATH))
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library does not export the declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    return handle


def last_error() -> str:
    msg = lib().rl_last_error()
    return msg.decode() if msg else ""


def check(status: int) -> None:
    if status == RL_OK:
        return
    msg = last_error() or f"libraglite_hip status {status}"
    if status == RL_ERR_UNSUPPORTED:
        raise UnsupportedError(msg)
    if status == RL_ERR_INVALID:
        raise ValueError(msg)
    if status == RL_ERR_NOMEM:
        raise MemoryError(msg)
    raise RagliteHipError(msg)


_initialised: set[int] = set()


def init(device: int = 0) -> None:
    """`hipSetDevice` for the calling thread + gfx950 check (idempotent per device)."""
    check(lib().rl_init(device))
    _initialised.add(device)


def device_count() -> int:
    n = c_int(0)
    check(lib().rl_device_count(C.byref(n)))
    return n.value


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(64)
    cus, mem = c_int(0), c_i64(0)
    check(lib().rl_device_info(device, name, 64, C.byref(cus), C.byref(mem)))
    return {"arch": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}
