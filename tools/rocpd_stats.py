"""Kernel statistics of a rocprofv3 --kernel-trace run, from its rocpd sqlite database (ROCm 7's default output format) or
from its *_kernel_trace.csv: per kernel name calls, total / average / min / max duration - the table `--stats` prints.

usage: python tools/rocpd_stats.py <results.db | dir-with-csv> <out.csv> [--launches]"""
import csv, glob, os, sqlite3, subprocess, sys
from collections import defaultdict


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True, text=True)
        return out.stdout.split("\n")[:len(names)]
    except Exception:
        return names


src, dst = sys.argv[1], sys.argv[2]
dur = defaultdict(list)
if src.endswith(".db"):
    c = sqlite3.connect(src)
    for name, d in c.execute("select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"):
        dur[name].append(d)
else:
    for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = list(dur)
pretty = dict(zip(names, demangle(names)))
total = sum(sum(v) for v in dur.values())
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([pretty[n], len(v), sum(v), f"{sum(v) / len(v):.1f}", f"{100.0 * sum(v) / total:.2f}", min(v), max(v)])
print(f"{sum(len(v) for v in dur.values())} launches, {total / 1e6:.2f} ms of kernel time -> {dst}")
