"""Per-site hardware counters of the GEMM launches of a bench.py run (developer tooling, not part of the product).

rocprofv3 --pmc serialises the dispatches of a process and numbers them in host enqueue order; bench.py --launch-log writes
the engine's GEMM launch log of the timed region in the same order.  This script takes the LAST len(log) GEMM dispatches of
every counter_collection.csv, checks that kernel template and grid agree with the log record they are matched to, and
averages every counter per (model site, kind).

    python tools/pmc_sites.py <launch_log.json> <out.json> <pass1 counter_collection.csv> [<pass2 ...> ...]

What the numbers are (MI355X_MICROARCH.md, rocprofv3 PMC section): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (= 16 x the number of 16x16x32 MFMAs);
SQ_BUSY_CYCLES is summed over the shader engines; FETCH_SIZE / WRITE_SIZE are in KiB and FETCH_SIZE is doubled on gfx950;
TCC_HIT / TCC_MISS are L2 (per-XCD) requests.  In-situ here means: the caches, clocks and launch order of the real step --
NOT concurrency: under --pmc one kernel runs at a time.
"""
import csv
import json
import re
import sys
from collections import defaultdict

GEMM = re.compile(r"gemm_(pipe|splitk|group|ldr|group_ldr|f8|f8t|f8t_group)?_?kernel<")


def kind_of_name(name):
    m = re.search(r"gemm_(pipe|splitk|group|ldr|group_ldr)_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+)", name)
    if m:
        return "wgrad" if m.group(6) == "true" else ("dgrad" if m.group(7) == "true" else "fwd")
    m = re.search(r"gemm_kernel<(\d+), (\d+), (true|false), (true|false)>", name)
    if m:
        return "wgrad" if m.group(3) == "true" else ("dgrad" if m.group(4) == "true" else "fwd")
    if "gemm_f8t" in name:
        return "wgrad"        # token-major fp8 weight gradients
    m = re.search(r"gemm_f8_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)[,>]", name)
    return "dgrad" if (m and m.group(6) == "true") else "fwd"          # gemm_f8_kernel: A_BF8 = an e5m2 gradient operand


def load(path):
    """{dispatch id: (kernel name, {counter: value})} of the GEMM dispatches of one pass."""
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if not GEMM.search(name):
                continue
            d = int(r["Dispatch_Id"])
            ent = rows.setdefault(d, (name, {}, int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)))
            ent[1][r["Counter_Name"]] = ent[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):       # kernel begin / end stamps of the (serialised) dispatch, ns
                ent[1]["_duration_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return [rows[d] for d in sorted(rows)]


def main():
    with open(sys.argv[1]) as f:
        log = json.load(f)
    recs = log["launches"]
    out_path, passes = sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    meta = {}
    mismatches = 0
    for path in passes:
        rows = load(path)
        if len(rows) < len(recs):
            raise SystemExit("%s: %d GEMM dispatches, the launch log has %d" % (path, len(rows), len(recs)))
        rows = rows[len(rows) - len(recs):]
        for rec, (name, ctr, grid, wg) in zip(recs, rows):
            if kind_of_name(name) != rec["kind"] or (("group" in name) != (rec["n_problems"] > 1)) or (("splitk" in name) != (rec["split_k"] > 1)):
                mismatches += 1
                continue
            key = "%s.%s" % (rec["site"], rec["kind"])
            meta[key] = dict(M=rec["M"], N=rec["N"], K=rec["K"], cfg=rec["cfg"], split_k=rec["split_k"], kernel=re.sub(r"\(anonymous namespace\)::", "", name)[:100],
                             gflop=rec["flops"] / 1e9, workgroups=grid // max(wg, 1))
            for c, v in ctr.items():
                acc[key][c] += v
                cnt[key][c] += 1
    sites = {}
    for key in sorted(acc):
        m = {c: acc[key][c] / cnt[key][c] for c in acc[key]}
        e = dict(meta[key], launches=max(cnt[key].values()), counters=m)
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            e["read_bytes_per_launch"] = 2.0 * m.get("FETCH_SIZE", 0.0) * 1024.0          # gfx950: x2 (MI355X_MICROARCH.md, HBM section)
            e["write_bytes_per_launch"] = m.get("WRITE_SIZE", 0.0) * 1024.0
            e["bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            e["l2_hit_rate"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0)
        if "SQ_WAVE_CYCLES" in m:
            wc = max(m["SQ_WAVE_CYCLES"], 1.0)
            e["wave_wait_frac"] = m.get("SQ_WAIT_ANY", 0.0) / wc
            e["wave_issue_stall_frac"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
        if "_duration_ns" in m:
            e["us_serialised"] = m.pop("_duration_ns") / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "us_serialised" in e:
            # MFMA-busy is summed over the 1024 SIMDs; the denominator is the kernel's own duration at the 2.4 GHz peak clock (the
            # chip holds less under load: a lower bound of the busy fraction).  GRBM_GUI_ACTIVE / 8 is NOT used as the duration:
            # on dispatches this short it reads 20+ us high (MI355X_MICROARCH.md, DVFS give-back)
            e["mfma_busy_frac_of_chip"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["us_serialised"] * 2400.0 * 1024.0)
            e["tflops_serialised"] = e["gflop"] / e["us_serialised"] * 1e3          # GFLOP / us = PFLOP/s
        sites[key] = e
    out = dict(_source_hash=log.get("source_hash"), _workload=log.get("workload"), _dtype=log.get("dtype", "bf16"), _steps=log.get("steps"), _mismatched_dispatches=mismatches,
               _note="per launch averages; dispatches serialised by the profiler (caches / order of the real step, no concurrency)", sites=sites)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print("matched %d launches per pass, %d mismatches" % (len(recs), mismatches))
    if mismatches:
        print("TABLE INVALID: the dispatches of at least one pass do not line up with the launch log (kernels launched after the logged "
              "region? dropped records?) -- bench.py ignores a table with _mismatched_dispatches > 0")
    hdr = "%-18s %5s %6s %6s %6s %4s %7s %7s %7s %7s %6s %6s %6s %6s" % ("site", "n", "M", "N", "K", "cfg", "GFLOP", "us(ser)", "rd MB", "wr MB", "L2hit", "wait", "stall", "mfma")
    print(hdr)
    for key, e in sites.items():
        print("%-18s %5d %6d %6d %6d %4d %7.2f %7.1f %7.2f %7.2f %6s %6s %6s %6s" % (
            key, e["launches"], e["M"], e["N"], e["K"], e["cfg"], e["gflop"], e.get("us_serialised", 0.0), e.get("read_bytes_per_launch", 0) / 1e6,
            e.get("write_bytes_per_launch", 0) / 1e6, ("%.3f" % e["l2_hit_rate"]) if "l2_hit_rate" in e else "-",
            ("%.3f" % e["wave_wait_frac"]) if "wave_wait_frac" in e else "-",
            ("%.3f" % e["wave_issue_stall_frac"]) if "wave_issue_stall_frac" in e else "-",
            ("%.3f" % e["mfma_busy_frac_of_chip"]) if "mfma_busy_frac_of_chip" in e else "-"))


if __name__ == "__main__":
    main()
