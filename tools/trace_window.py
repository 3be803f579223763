"""Every kernel of a time window of ONE step from a rocprofv3 --kernel-trace CSV, all queues, in start order (developer tooling).

usage: python tools/trace_window.py <kernel_trace.csv> <from_ms> <to_ms>     (times relative to the step's first kernel)
"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:52]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "adamw" in r[2]]
bounds = [m for j, m in enumerate(marks) if j == 0 or marks[j - 1] != m - 1 and rows[m][0] - rows[marks[j - 1]][1] > 2_000_000]
a, b = bounds[-3], bounds[-2]
step = rows[a:b]
t0 = step[0][0]
lo, hi = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
queues = sorted({(r[3], r[4]) for r in step})
col = {q: i for i, q in enumerate(queues)}
last_end = {q: None for q in queues}
print("queues: " + "  ".join("%d=%s" % (i, q) for q, i in col.items()))
for r in step:
    q = (r[3], r[4])
    gap = (r[0] - last_end[q]) / 1e3 if last_end[q] is not None else 0.0
    last_end[q] = r[1]
    if lo <= r[0] - t0 <= hi:
        print("%8.1f us  q%d %s%-52s %6.1f us   (queue idle before: %6.1f us)" % ((r[0] - t0) / 1e3, col[q], "    " * col[q], short(r[2]), (r[1] - r[0]) / 1e3, gap))
