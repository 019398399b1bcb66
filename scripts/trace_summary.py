"""Summarise a GEMMTRACE dump (stderr of a RAGLITE_GEMM_TRACE=1 run): mean cycles per slab and wave for each phase.
Columns: before-vmcnt after-vmcnt after-barrier after-pair0..7 after-advance."""
import re
import sys
from collections import defaultdict

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"GEMMTRACE slab (\d+) wave (\d+):(.*)", line)
    if m:
        rows[(int(m.group(1)), int(m.group(2)))] = [int(x) for x in m.group(3).split()]
slabs = sorted({g for g, _ in rows})
names = ["vmcnt", "barrier"] + [f"p{i}" for i in range(8)] + ["advance", "next"]
acc = defaultdict(lambda: defaultdict(list))
for g in slabs[:-1]:
    for w in range(8):
        r, nxt = rows[(g, w)], rows[(g + 1, w)]
        d = [r[i + 1] - r[i] for i in range(11)] + [nxt[0] - r[11]]
        for n, v in zip(names, d):
            acc[w][n].append(v)
        acc[w]["period"].append(nxt[0] - r[0])
for w in range(8):
    print(f"wave {w}: " + " ".join(f"{n}={sum(v)/max(1,len(v)):5.0f}" for n, v in acc[w].items()))
