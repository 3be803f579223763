"""Developer helper: tabulate tools/gemm_lab.bin outputs (us per launch) as shape x configuration.
    python tools/lab_table.py <tile,tile,...> file [file ...]"""
import collections
import re
import sys


def load(p, d):
    for l in open(p):
        m = re.match(r"(.{12}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) tile=\s*(\d+) S=(\d)\s+([\d.]+) us\s+([\d.]+) TF\s+maxdiff (\S+)(.*)", l)
        if m and m.group(6) == '1':
            d.setdefault(m.group(1).strip(), {})[int(m.group(5))] = (float(m.group(7)), 'MISMATCH' in l)
    return d


tiles = [int(t) for t in sys.argv[1].split(",")]
d = collections.OrderedDict()
for f in sys.argv[2:]:
    load(f, d)
print("us".ljust(14) + "".join("%8d" % t for t in tiles))
for name, row in d.items():
    print(name.ljust(14) + "".join(("%7.1f%s" % (row[t][0], '!' if row[t][1] else ' ')) if t in row else "       -" for t in tiles))
