"""Print the kernel timeline (start offset, duration, gap to the previous kernel; us) of the last few dispatches of a
rocprofv3 --kernel-trace CSV.  Usage: python scripts/trace_timeline.py <kernel_trace.csv> [n_last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
