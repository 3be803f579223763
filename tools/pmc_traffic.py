"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-kernel HBM traffic per launch.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B
by rocprofv3's derived metrics; on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B / lane) reads at
64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            a = acc[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        nf, vf = fetch.get(k, [0, 0.0])
        nw, vw = write.get(k, [0, 0.0])
        rd = 2.0 * vf * 1024.0 / max(nf, 1)          # gfx950 correction: x2
        wr = vw * 1024.0 / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr}
        m = re.match(r"void gemm_(pipe|group)_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+)>", k)
        if m:      # the label bench.py prints for this GEMM variant
            tm, tn, wm, wn, ns = int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(8))
            var = "wgrad" if m.group(6) == "true" else ("dgrad" if m.group(7) == "true" else "fwd")
            out[k]["bench_label"] = "gemm<dma%dx%dw%ds%d,%s>" % (32 * tm, 32 * tn, wm * wn, ns, var)
    # stamp: bench.py reports these numbers only while the kernel sources still hash to what was measured
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ("cqa-crct_amd/csrc/gemm.hip", "cqa-crct_amd/csrc/engine.cpp"):       # = bench.KERNEL_SOURCES
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    out["_source_hash"] = {"value": h.hexdigest()[:16], "files": ["cqa-crct_amd/csrc/gemm.hip", "cqa-crct_amd/csrc/engine.cpp"]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in [kv for kv in out.items() if not kv[0].startswith("_")][:25]:
        print("%-90s n=%5d  rd %8.2f MB  wr %8.2f MB" % (k[:90], v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
