"""A readable digest of one bench.py line (the GPU-call scripts tee it into their summary.txt).  python scripts/bench_summary.py bench.json"""
import json
import sys

try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = r["roofline"]
    print(f"  {r['value']:.0f} q/s  {r['ms_per_step']:.3f} ms/step  launch {rf['kernel_ms']:.4f} ms ({rf.get('passes_per_launch')} passes: {rf.get('kernel_ms_per_pass', float('nan')):.4f} ms per pass) "
          f"frac {rf['frac']:.3f} recall {r.get('recall_at_100')} cand {r.get('candidates_per_query')} fb {r.get('fallback_steps')}")
    print("  sustained", {k: rf.get(k) for k in ("sustained_tflops", "frac_of_sustained", "shader_clock_ghz")})
    print("  vendor", {k: v for k, v in rf.items() if k.startswith("vendor_gemm") and k != "vendor_gemm_how"}, {k: rf.get(k) for k in ("pass_kernel_over_vendor", "pass_kernel_over_vendor_square")})
    print("  traffic", rf.get("traffic"), "|", (rf.get("traffic_source") or "")[:120])
    print("  memory", {k: v for k, v in (r.get("index_memory") or {}).items() if k != "note"})
    if r.get("exact_fp32"):
        print("  exact_fp32", {k: r["exact_fp32"].get(k) for k in ("value", "ms_per_step", "frac")})
    if r.get("f16_stored"):
        print("  f16_stored", {k: r["f16_stored"].get(k) for k in ("value", "ms_per_step", "candidates_per_query")})
    if r.get("f16_queries"):
        print("  f16_queries", {k: r["f16_queries"].get(k) for k in ("value", "ms_per_step", "route", "fallback", "score_max_rel_err", "recall_at_100_slab")})
    for k, v in (r.get("configs") or {}).items():
        roof = v.get("roofline") or {}
        print("  ", k, {kk: v.get(kk) for kk in ("value", "ms_per_query", "ms", "ms_per_batch", "ms_per_launch", "error") if v.get(kk) is not None},
              {kk: roof.get(kk) for kk in ("frac", "kernel_frac", "frac_of_hbm_from_counters") if roof.get(kk) is not None}, v.get("check") if k.startswith("cfg5_full") else "")
    for k, v in (r.get("raglite_shaped") or {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "candidates_per_query", "fallback_steps", "error")}, "f16 queries:",
              {kk: (v.get("f16_queries") or {}).get(kk) for kk in ("value", "route")})
    print("  tol", r.get("score_tolerance"))
    print("  cpu", r.get("cpu_baseline", {}).get("value"), r.get("cpu_baseline", {}).get("cores"), "| fraction_check:", r.get("fraction_check"))
except Exception as exc:  # noqa: BLE001
    print("  (no bench line)", exc)
