"""Per-queue timeline of ONE training step from a rocprofv3 --kernel-trace CSV.

usage: python tools/trace_timeline.py <kernel_trace.csv> [out.txt]
Steps are delimited by the last adamw_kernel launches of consecutive steps.  For every hardware queue the
script prints busy time, and for the queue that carries most work the dependent-launch gaps.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                         r.get("Stream_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    # step boundary = first adamw launch after a gap of non-adamw kernels
    bounds = [m for j, m in enumerate(marks) if j == 0 or marks[j - 1] != m - 1 and rows[m][0] - rows[marks[j - 1]][1] > 2_000_000]
    if len(bounds) < 3:
        print("not enough steps", len(bounds))
        return
    a, b = bounds[-3], bounds[-2]
    step = rows[a:b]
    t0 = step[0][0]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    print("step span %.3f ms, %d kernels" % ((step[-1][1] - t0) / 1e6, len(step)), file=out)
    byq = defaultdict(list)
    for r in step:
        byq[(r[3], r[4])].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[1] - r[0] for r in rs)
        print("queue %s: %d kernels, busy %.3f ms, first %.3f last %.3f" % (q, len(rs), busy / 1e6, (rs[0][0] - t0) / 1e6,
                                                                           (rs[-1][1] - t0) / 1e6), file=out)
    # union busy (any queue running)
    ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    depth, last, hist = 0, t0, defaultdict(int)
    for t, d in ev:
        hist[depth] += t - last
        last, depth = t, depth + d
    print("time with k kernels in flight: " + ", ".join("%d: %.3f ms" % (k, v / 1e6) for k, v in sorted(hist.items())), file=out)
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        print("\n== queue %s" % (q,), file=out)
        gap_by = defaultdict(lambda: [0, 0, 0])
        prev = None
        for r in rs:
            gap = (r[0] - prev) if prev is not None else 0
            g = gap_by[short(r[2])]
            g[0] += 1; g[1] += r[1] - r[0]; g[2] += max(gap, 0)
            prev = r[1]
        for k, (n, dur, gap) in sorted(gap_by.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print("  %-70s n=%3d dur %8.1f us (avg %6.1f)  gap-before %8.1f us (avg %5.1f)" % (k, n, dur / 1e3, dur / 1e3 / n, gap / 1e3,
                                                                                              gap / 1e3 / n), file=out)
    print("\n== full sequence (t_start us, queue, dur us, name)", file=out)
    for r in step:
        print("%9.1f q%s/%s %7.1f %s" % ((r[0] - t0) / 1e3, r[3], r[4], (r[1] - r[0]) / 1e3, short(r[2])), file=out)


if __name__ == "__main__":
    main()
