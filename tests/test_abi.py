"""The C-ABI library loads on a CPU-only box and exports exactly what include/raglite_hip.h declares.
Only argument validation that returns before the first HIP call is exercised here."""

import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from raglite_amd import _abi

HEADER = Path(__file__).resolve().parent.parent / "include" / "raglite_hip.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    declared = _declared()
    assert len(declared) >= 20
    handle = C.CDLL(str(_abi.LIB_PATH))
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in raglite_hip.h but not exported"
    assert sorted(_abi.EXPORTED_SYMBOLS) == declared, "ctypes signatures out of sync with the header"


def test_version_and_error_channel():
    lib = _abi.lib()
    assert lib.rl_version() == 100
    # dim <= 0 is rejected before any HIP call
    st = lib.rl_pool_norm(None, 0, 0, None, None, 0, 1, 0.0, None, None, _abi.MEM_HOST, None)
    assert st == _abi.RL_ERR_INVALID
    assert "rl_pool_norm" in _abi.last_error()
    with pytest.raises(ValueError, match="rl_pool_norm"):
        _abi.check(st)


def test_argument_validation_without_gpu():
    lib = _abi.lib()
    tok = np.zeros((4, 8), dtype=np.float32)
    b = np.array([0, 3], dtype=np.int64)
    e = np.array([2, 9], dtype=np.int64)  # 9 > 4 rows
    out = np.zeros((2, 8), dtype=np.float32)
    st = lib.rl_pool_norm(tok.ctypes.data, 4, 8, b.ctypes.data, e.ctypes.data, 2, 1, 0.0, out.ctypes.data, None,
                          _abi.MEM_HOST, None)
    assert st == _abi.RL_ERR_INVALID and "span" in _abi.last_error()
    st = lib.rl_pool_norm(tok.ctypes.data, 4, 8, b.ctypes.data, e.ctypes.data, 2, 1, 0.0, None, None,
                          _abi.MEM_HOST, None)
    assert st == _abi.RL_ERR_INVALID and "no output" in _abi.last_error()
    h = C.c_void_p()
    off = np.array([0, 2, 1, 4], dtype=np.int64)  # not ascending
    st = lib.rl_index_create(C.byref(h), tok.ctypes.data, 4, 8, off.ctypes.data, 3, 0, _abi.MEM_HOST, None)
    assert st == _abi.RL_ERR_INVALID and "ascending" in _abi.last_error() and not h.value
    st = lib.rl_index_create(C.byref(h), tok.ctypes.data, 4, 8, None, 0, 7, _abi.MEM_HOST, None)
    assert st == _abi.RL_ERR_INVALID and "metric" in _abi.last_error()
    assert lib.rl_search_rows(None, None, 1, 10, None, None, 0, None) == _abi.RL_ERR_INVALID
    assert lib.rl_topk(None, 1, 10, 10, 4096, None, None, 0, None) == _abi.RL_ERR_UNSUPPORTED
    assert lib.rl_merge_topk(None, None, 0, 1, 1, 1, None, None, 0, None) == _abi.RL_ERR_INVALID


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU the ops raise instead of computing elsewhere."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import raglite_amd

    with pytest.raises(RuntimeError):
        raglite_amd.pool_norm(np.zeros((4, 8), np.float32), np.array([0]), np.array([4]))
    with pytest.raises(RuntimeError):
        raglite_amd.DeviceIndex(np.zeros((4, 8), np.float32))


def test_no_oracle_import_in_product():
    """Parity rule: nothing under raglite_amd/ may import oracle/."""
    root = Path(__file__).resolve().parent.parent / "raglite_amd"
    for path in root.rglob("*.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle\b", path.read_text(), flags=re.M), path


def test_graft_entry_build_runs():
    """The driver's per-round build check: compiles (incrementally) every HIP source, binds the ABI and, where the
    reference is mounted, regenerates every golden fixture from the reference's own code and checks nothing drifted."""
    import __graft_entry__

    __graft_entry__.build()


def test_c_consumers_compile_against_the_header_alone():
    """include/raglite_hip.h is the boundary: plain C11, no HIP / torch headers.  The pure-C consumers (the GPU smoke of
    tests/test_gpu_parity.py::test_pure_c_consumer and the measurement probe scripts/micro/r3_probe.c) must compile against it
    with warnings as errors -- checked here without a GPU (syntax + types only)."""
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    for src in (root / "tests" / "c" / "abi_smoke.c", root / "scripts" / "micro" / "r3_probe.c"):
        res = subprocess.run([gcc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", f"-I{root / 'include'}", str(src)],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr


def test_shipped_library_reads_no_environment_variable():
    """Route switches are options of an index (`rl_index_set_option`); timing-experiment builds of the kernels exist only in
    libraglite_hip_exp.so.  The shipped library does not even import getenv, and carries no RAGLITE_* string."""
    import subprocess

    syms = subprocess.run(["nm", "-D", "--undefined-only", str(_abi.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms, [ln for ln in syms.splitlines() if "getenv" in ln]
    blob = Path(_abi.LIB_PATH).read_bytes()
    assert b"RAGLITE_" not in blob


def test_default_options_without_gpu():
    """`rl_set_default_option` / `rl_get_default_option` touch no device: defaults, validation, round trip."""
    lib = _abi.lib()
    v = C.c_int64(-7)
    defaults = {"hi_search": 1, "hi_maxsim": 1, "hi_products": 1, "pp_pass": 1, "fused_topk": 1, "fused_hi": 1, "fused_pp": 1,
                "fused_topk_cap": 0, "fused_topk_stride": 0, "gemm_pass": 1, "query_pairs": 1, "planes_gemm": 1, "keep_image": 1,
                "keep_hi": 1, "image_headroom_mb": -1, "arithmetic": 0, "exact_kth_threshold": 1, "fused_two_rounds": 1,
                "keep_hi_plane": 1, "pairs_packed": 2, "f16_exact": 1, "lazy_images": 1, "fused_pp_sample": 1, "list_select": 1, "hi_few": 1, "topk_block": 2, "hi_pivot": 1}
    assert set(defaults) == set(_abi.OPTIONS)
    for name, want in defaults.items():
        assert lib.rl_get_default_option(_abi.OPTIONS[name], C.byref(v)) == _abi.RL_OK and v.value == want, name
    for key, value in ((0, 1), (28, 1), (99, 0), (_abi.OPTIONS["hi_products"], 3), (_abi.OPTIONS["hi_search"], 2),
                       (_abi.OPTIONS["fused_topk_cap"], 8193), (_abi.OPTIONS["fused_topk_stride"], 1), (_abi.OPTIONS["arithmetic"], 2),
                       (_abi.OPTIONS["image_headroom_mb"], -2)):
        assert lib.rl_set_default_option(key, value) == _abi.RL_ERR_INVALID, (key, value)
    assert "rl_set_default_option" in _abi.last_error()
    assert lib.rl_set_default_option(_abi.OPTIONS["fused_topk_cap"], 64) == _abi.RL_OK
    assert lib.rl_get_default_option(_abi.OPTIONS["fused_topk_cap"], C.byref(v)) == _abi.RL_OK and v.value == 64
    assert lib.rl_set_default_option(_abi.OPTIONS["fused_topk_cap"], 0) == _abi.RL_OK
    assert lib.rl_index_set_option(None, 1, 1) == _abi.RL_ERR_INVALID and lib.rl_index_get_option(None, 1, C.byref(v)) == _abi.RL_ERR_INVALID


def test_option_keys_match_the_header_enum():
    """`raglite_amd._abi.OPTIONS` is the Python spelling of `enum rl_option` in include/raglite_hip.h: same names, same numbers, and every
    key the header declares is documented in its options table."""
    import re
    from pathlib import Path

    from raglite_amd import _abi

    text = (Path(__file__).resolve().parent.parent / "include" / "raglite_hip.h").read_text()
    end = text.index("} rl_option;")
    start = text.rindex("enum", 0, end)
    body = text[start:end]
    pairs = {m.group(1): int(m.group(2)) for m in re.finditer(r"RL_OPT_([A-Z0-9_]+)\s*=\s*(\d+)", body)}
    count = pairs.pop("COUNT_")
    assert count == max(pairs.values()) + 1 and sorted(pairs.values()) == list(range(1, count))
    assert {name.lower(): key for name, key in pairs.items()} == _abi.OPTIONS
    table = text[:start]
    for name in pairs:
        assert f"RL_OPT_{name} " in table or f"RL_OPT_{name}\n" in table, f"RL_OPT_{name} is not in the header's options table"
