"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name the per-launch durations of the LAST step.

usage: python tools/trace_summary.py <dir-with-*_kernel_trace.csv> [launches_per_step_hint]
"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("ttk::", "").replace("(anonymous namespace)::", "")
    return n[:70]
by = defaultdict(list)
for s, e, n in rows:
    by[short(n)].append((e - s) / 1000.0)
tot = sum(sum(v) for v in by.values())
print(f"{len(rows)} launches, total {tot/1000:.2f} ms")
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    per = len(v) // steps if steps > 1 else len(v)
    last = v[-per:] if per else v
    print(f"{sum(v)/max(steps,1):9.1f} us/step  n={per:4d}  {n}")
    if per <= 40 and "-v" in sys.argv:
        print("           ", " ".join(f"{x:.0f}" for x in last))
