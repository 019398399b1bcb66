"""Instruction stream of one kernel out of a hipcc -S listing, labels and comments stripped: what two builds of the same kernel are
compared by (python scripts/dev/kernel_stream.py listing.s 'name regex' > stream.txt)."""
import re
import sys

listing, pat = sys.argv[1], re.compile(sys.argv[2])
out, on = [], False
for line in open(listing):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        on = bool(pat.search(m.group(1)))
        continue
    if not on:
        continue
    t = re.sub(r";.*", "", line).strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    t = re.sub(r"\.LBB\d+_\d+", "LBL", t)
    out.append(t)
    if t == "s_endpgm":
        on = False
print("\n".join(out))
