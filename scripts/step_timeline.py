"""Where a step's time goes, launch by launch, from a rocprofv3 --kernel-trace CSV: duration of every kernel of one period and the idle gap in
front of it, averaged over the periods that have the most common launch count.

    python scripts/step_timeline.py <kernel_trace.csv> <first-kernel substring> [<must-contain substring>] [--tail N]

A period starts at a launch whose name contains the first substring and whose predecessor's does not; periods that lack the second substring
are dropped (other blocks of the same program that start with the same kernel).  --tail N: no periods, the last N launches as they are."""
import csv
import sys
from collections import Counter

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = list(csv.DictReader(open(args[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(r):
    n = r["Kernel_Name"]
    n = n.replace("rl::(anonymous namespace)::", "").replace("rl::", "").replace("void ", "")
    return n.split("(")[0][:64]


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


if "--tail" in sys.argv:
    n = int(sys.argv[sys.argv.index("--tail") + 1])
    first = max(1, len(rows) - n)
    for j in range(first, len(rows)):
        gap = (int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"])) / 1e3
        print(f"{j - first:3d} {short(rows[j]):64s} dur {dur(rows[j]):8.1f} us   gap before {gap:7.1f} us")
    sys.exit(0)

mark = args[1]
need = args[2] if len(args) > 2 else None
starts = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"] and (i == 0 or mark not in rows[i - 1]["Kernel_Name"])]
periods = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if need is None or any(need in rows[j]["Kernel_Name"] for j in range(a, b))]
if not periods:
    sys.exit("no period found")
modal = Counter(b - a for a, b in periods).most_common(1)[0][0]
periods = [(a, b) for a, b in periods if b - a == modal][1:]  # (the first one carries warm-up effects)
wall = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in periods]
print(f"{len(periods)} periods of {modal} launches; period (us): mean {sum(wall) / len(wall):.1f} min {min(wall):.1f} max {max(wall):.1f}")
sd = sg = 0.0
for pos in range(modal):
    ds = [dur(rows[a + pos]) for a, _ in periods]
    gs = [(int(rows[a + pos]["Start_Timestamp"]) - int(rows[a + pos - 1]["End_Timestamp"])) / 1e3 for a, _ in periods if a + pos > 0]
    d, g = sum(ds) / len(ds), (sum(gs) / len(gs) if gs else 0.0)
    sd += d
    sg += g
    print(f"{pos:3d} {short(rows[periods[0][0] + pos]):64s} dur {d:8.1f} us   gap before {g:7.1f} us")
print(f"sum of durations {sd:.1f} us, sum of gaps {sg:.1f} us")
