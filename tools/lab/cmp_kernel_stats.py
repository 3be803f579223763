"""Lab: two `rocprofv3 --kernel-trace --stats --output-format csv -d <dir>` runs side by side, per kernel: calls, average duration in each,
difference of the totals -- largest differences first.  Used for the price of the fp32 residual stream per kernel
(profiles/r6_residual_stream_parity.txt):   python tools/lab/cmp_kernel_stats.py <dir of run A> <dir of run B> [name filter]"""
import csv, sys, glob
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load(sys.argv[1]), load(sys.argv[2])
print("%-92s %6s %8s %8s %9s" % ("kernel", "calls", "A us", "B us", "d total ms"))
rows = []
for k in set(a) | set(b):
    ca, ua, ta = a.get(k, (0, 0, 0)); cb, ub, tb = b.get(k, (0, 0, 0))
    rows.append((ta - tb, k, ca, ua, ub))
pick = sys.argv[3] if len(sys.argv) > 3 else None      # optional third argument: only kernels whose name contains it (all of them)
rows = [r for r in rows if pick in r[1]] if pick else sorted(rows, key=lambda r: -abs(r[0]))[:16]
for d, k, c, ua, ub in sorted(rows, key=lambda r: -abs(r[0])):
    print("%-92s %6d %8.1f %8.1f %9.2f" % (k[:92], c, ua, ub, d))
