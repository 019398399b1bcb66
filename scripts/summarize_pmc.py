"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

out = Path(sys.argv[1])
for f in sorted(out.rglob("*counter_collection*.csv")):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")[:70]
            acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
    print(f"--- {f}")
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k:70s} {c:28s} n={len(v):5d} mean={sum(v)/len(v):.6g}")
