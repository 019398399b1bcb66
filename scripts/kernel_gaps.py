"""Per-kernel duration and the idle gap in front of each launch, from a rocprofv3 --kernel-trace CSV: the launch-bound tail of a
single-query search (BASELINE cfg 2).  python scripts/kernel_gaps.py <kernel_trace.csv> [first_kernel_substring]"""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "maxsim_stream_kernel"
# split into queries: a query starts at each launch whose name contains `mark` and follows a non-`mark` launch pattern start
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# keep the launches of the last 5 complete periods between consecutive "first" marks: the first kernel of a query is the first `mark`
# launch after a launch of "merge_topk" or "topk_final" (the end of the previous query)
qstart = [i for i in starts if i == 0 or any(t in rows[i - 1]["Kernel_Name"] for t in ("topk_final", "merge_topk", "topk_filter"))]
if len(qstart) < 4:
    qstart = starts
periods = list(zip(qstart[-6:-1], qstart[-5:]))
agg = OrderedDict()
total = []
for a, b in periods:
    total.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
    for j in range(a, b):
        r = rows[j]
        name = r["Kernel_Name"].split("(")[0][-48:]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3 if j > 0 else 0.0
        key = (j - a, name)
        agg.setdefault(key, []).append((dur, gap))
print(f"periods (us): {[round(t, 1) for t in total]}")
sd = sg = 0.0
for (pos, name), v in agg.items():
    d = sum(x[0] for x in v) / len(v)
    g = sum(x[1] for x in v) / len(v)
    sd += d
    sg += g
    print(f"{pos:3d} {name:50s} dur {d:8.1f} us   gap before {g:6.1f} us   (n={len(v)})")
print(f"sum of durations {sd:.1f} us, sum of gaps {sg:.1f} us")
