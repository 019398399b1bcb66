"""bench.py's line, checked without a GPU: every printed roofline fraction lies in (0, 1] (round 4's driver record carried
`exact_fp32.frac` = 6.08: a per-launch byte count of an 8-pass launch reused for a one-pass kernel), every HBM-bound block's
algorithmic bytes / kernel time stays under the HBM peak unless the block names the narrower image it streams, and every kernel
name the bench prints is a kernel the built library really holds (round 5 changed a template list and the old name stayed)."""

from __future__ import annotations

import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def test_checker_flags_the_round_4_line():
    """The driver's round-4 record, as it was printed: the checker must name exact_fp32.frac."""
    rec = ROOT / "BENCH_r04.json"
    if not rec.exists():
        pytest.skip("no BENCH_r04.json in this checkout")
    m = re.search(r'"exact_fp32": (\{[^{}]*\})', json.loads(rec.read_text())["run"]["stdout_tail"])  # (the driver keeps the line's tail)
    assert m, "the record's tail no longer holds the exact_fp32 block"
    bad = bench.fraction_violations({"exact_fp32": json.loads(m.group(1))})
    assert any(b.startswith("exact_fp32.frac") for b in bad), bad


def test_checker_rules():
    ok = {"roofline": {"bound": "mfma", "frac": 0.47, "hbm": {"frac": 0.28}}, "x": [{"kernel_frac": 1.0}]}
    assert bench.fraction_violations(ok) == []
    assert bench.fraction_violations({"a": {"frac": 6.08}}) == ["a.frac = 6.08 is not in (0, 1]"]
    assert bench.fraction_violations({"a": {"hbm_frac": 0.0}}) and bench.fraction_violations({"a": {"frac_of_sustained": float("nan")}})
    # an HBM-bound block whose algorithmic bytes / kernel time beats the HBM peak must say which narrower image it streams
    blk = {"bound": "hbm", "algorithmic_bytes": 4.096e9, "kernel_ms": 0.32, "frac": 0.8}
    assert len(bench.fraction_violations({"cfg2": {"roofline": blk}})) == 1
    named = dict(blk, narrower_image="fp16 HI plane, 2 B per element", frac_vs_4B_per_element_whole_query=1.38)
    assert bench.fraction_violations({"cfg2": {"roofline": named}}) == []
    # ... and only that one key may exceed 1 there
    assert bench.fraction_violations({"cfg2": {"roofline": dict(named, frac=1.2)}})


def _recorded_lines():
    out = []
    for path in sorted((ROOT / "profiles").glob("r*bench*.json")) + sorted(ROOT.glob("BENCH_r*.json")):
        try:
            doc = json.loads(path.read_text())
        except ValueError:
            continue
        doc = doc.get("parsed", doc) if isinstance(doc, dict) else None
        if isinstance(doc, dict) and doc.get("bench_schema", 0) >= bench.BENCH_SCHEMA:
            out.append((path.name, doc))
    return out


def test_recorded_lines_of_this_schema_have_sane_fractions():
    lines = _recorded_lines()
    if not lines:
        pytest.skip("no bench line of the current schema is recorded under profiles/ yet")
    for name, doc in lines:
        assert bench.fraction_violations(doc) == [], name
        assert str(doc.get("fraction_check", "")).startswith("ok"), name
        roof = doc["roofline"]
        assert 0.0 < roof["frac"] <= 1.0
        if doc.get("exact_fp32"):
            ex = doc["exact_fp32"]
            # one query per pass, one pass per launch: the fraction IS 4 N d bytes over the kernel time
            assert ex["frac"] == pytest.approx(ex["algorithmic_bytes_per_launch"] / (ex["kernel_ms"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, rel=1e-9), name
        # round 6: what the review's arguments lean on must be SCALARS where the driver keeps them (its record drops nested objects)
        for key in ("sustained_tflops", "frac_of_sustained", "shader_clock_ghz"):
            assert isinstance(roof.get(key), float), f"{name}: roofline.{key}"
        if doc.get("n_gpus") == 1 and "vendor_gemm_error" not in roof and "vendor_gemm_how" in roof:  # absent: run with --no-vendor-gemm
            for key in ("vendor_gemm_tflops", "pass_kernel_over_vendor"):
                assert isinstance(roof.get(key), float), f"{name}: roofline.{key}"
        assert isinstance(doc.get("index_memory_times_corpus"), float), name
        if doc.get("f16_queries"):
            assert isinstance(doc.get("f16_queries_value"), float), name
        if roof.get("traffic") is not None:  # only ever from a PMC record filed under the kernel the block reports
            assert roof["kernel"].split(" (")[0] in roof["traffic_source"], name


def _library_kernels() -> set[str]:
    lib = ROOT / "raglite_amd" / "_lib" / "libraglite_hip.so"
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not lib.exists() or not Path(filt).exists() or not shutil.which("strings"):
        pytest.skip("library, strings or c++filt not available")
    # the host side registers every kernel under its mangled name: the strings of the library hold them
    raw = subprocess.run(["strings", "-n", "8", str(lib)], capture_output=True, text=True, check=True).stdout
    mangled = sorted({w for w in raw.split() if w.startswith("_ZN2rl") and "kernel" in w})
    names = subprocess.run([filt], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.splitlines()
    return {re.sub(r"^void ", "", n).split("(")[0] for n in names}


def test_kernel_names_printed_by_the_bench_exist_in_the_library():
    have = _library_kernels()
    assert any("maxsim_pp_kernel" in h for h in have)
    wanted = set()
    for src in (ROOT / "bench.py", ROOT / "scripts" / "bench_configs.py"):
        for m in re.finditer(r'"((?:rl::)?[a-z_0-9]+_kernel<[^">]*>)', src.read_text()):
            wanted.add(m.group(1))
    assert len(wanted) >= 8
    norm = lambda s: re.sub(r"\s+", "", s if s.startswith("rl::") else "rl::" + s)  # noqa: E731
    have_n = {norm(re.sub(r"\(anonymous namespace\)::", "", h)) for h in have}
    missing = sorted(w for w in wanted if norm(w) not in have_n)
    assert not missing, f"bench prints kernel names the library does not hold: {missing}"
