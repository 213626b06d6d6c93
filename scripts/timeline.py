"""Prints the launch timeline of one timed bench step from a rocprofv3 kernel trace (csv): per kernel the start offset
from the step's first kernel, the duration and the queue it ran on."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("l3d::", "")[:34]
rows.sort(key=lambda r: r["s"])
# steps begin with the Steger mark kernel; take the third timed step (index from the end: profile pass follows)
marks = [i for i, r in enumerate(rows) if r["n"].startswith("k_steger_mark")]
pick = marks[4] if len(marks) > 6 else marks[len(marks) // 2]
nxt = marks[marks.index(pick) + 1]
t0 = rows[pick]["s"]
for r in rows[pick:nxt]:
    print(f"{(r['s'] - t0) / 1e6:8.3f} ms  +{(r['e'] - r['s']) / 1e6:7.3f} ms  q{r['Queue_Id']:>3}  {r['n']}")
print(f"step span {(max(r['e'] for r in rows[pick:nxt]) - t0) / 1e6:.3f} ms; next step starts at {(rows[nxt]['s'] - t0) / 1e6:.3f} ms")
