"""Print the kernel launch sequence of the LAST step in a rocprofv3 --kernel-trace CSV (steps are delimited by grad_sqnorm_k).

usage: python tools/trace_sequence.py <dir-with-*_kernel_trace.csv>
"""
import csv, glob, os, re, sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if "grad_sqnorm_k" in r[2]]
lo, hi = [(a + 1, b + 1) for a, b in zip(ends, ends[1:]) if b - a > 50][-1]  # skip optimizer-only timing calls
prev_end = rows[lo - 1][1]
for s, e, n in rows[lo:hi]:
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("ttk::", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    print(f"gap {(s - prev_end) / 1000.0:7.1f}  dur {(e - s) / 1000.0:7.1f}  {n[:100]}")
    prev_end = max(prev_end, e)
print(f"step span {(rows[hi - 1][1] - rows[lo][0]) / 1e6:.3f} ms, {hi - lo} launches")
