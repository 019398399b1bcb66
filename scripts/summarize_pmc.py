"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.

    python scripts/summarize_pmc.py OUT_DIR
    python scripts/summarize_pmc.py OUT_DIR --traffic-json profiles/traffic.json --passes-per-launch 8 --source profiles/r05_x_pmc_summary.txt

The second form also files the mean FETCH_SIZE of every MaxSim / search kernel under its own name (up to the argument list) in the
"by_kernel" table of profiles/traffic.json: bytes_per_launch = mean KiB x 1024 x 2 (gfx950 counts wide coalesced streams at half,
MI355X_MICROARCH.md section HBM).  bench.py prints `roofline.traffic` only from the record whose key equals `roofline.kernel`."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

args = sys.argv[1:]
out = Path(args[0])
opts = dict(zip(args[1::2], args[2::2]))
fetch = {}
for f in sorted(out.rglob("*counter_collection*.csv")):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print(f"--- {f}")
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k[:70]:70s} {c:28s} n={len(v):5d} mean={sum(v)/len(v):.6g}")
            if c == "FETCH_SIZE":
                fetch[k] = (sum(v) / len(v), len(v))

if "--traffic-json" in opts:
    path = Path(opts["--traffic-json"])
    tj = json.loads(path.read_text()) if path.exists() else {}
    table = tj.setdefault("by_kernel", {})
    for k, (mean_kib, n) in fetch.items():
        m = re.match(r"^(?:void )?(rl::[\w:]*(?:maxsim|scan|stream|pool_norm)\w*(?:<[^(]*>)?)\(", k)
        if not m:
            continue
        table[m.group(1)] = {"fetch_kib_mean": mean_kib, "bytes_per_launch": mean_kib * 1024 * 2, "dispatches": n,
                             # (the batch pipeline launches all passes of a step as grid rows of ONE launch of the sixteen-query kernel)
                             "passes_per_launch": int(opts.get("--passes-per-launch", 1)) if "maxsim_pp_kernel<0, 0" in m.group(1) else 1, "source": opts.get("--source", str(out))}
    # --workload KEY=KERNEL_SUBSTRING [--workload2 ...]: one kernel serves several workloads (cfg 3 at two pool sizes) -- file the record under
    # by_workload[KEY] with the kernel's name, so that the bench block of THAT workload can pick it up and no other
    for opt, val in opts.items():
        if not opt.startswith("--workload"):
            continue
        wkey, needle = val.split("=", 1)
        for k, (mean_kib, n) in fetch.items():
            m = re.match(r"^(?:void )?(rl::[\w:]*\w+(?:<[^(]*>)?)\(", k)
            if m and needle in m.group(1):
                tj.setdefault("by_workload", {})[wkey] = {"kernel": m.group(1), "fetch_kib_mean": mean_kib, "bytes_per_launch": mean_kib * 1024 * 2,
                                                          "dispatches": n, "source": opts.get("--source", str(out))}
    path.write_text(json.dumps(tj, indent=2) + "\n")
