"""In-step sweep of the GEMM shape-class table (bench.py --class-policy) and of the grouped weight-gradient configuration
(--wgrad-cfg) for one workload, every variant a fresh bench.py process on the same box; the first and last rows are the
unmodified step (box drift).  Usage on the GPU box:

    python tools/lab/class_sweep.py --workload plotqa-real "L.w=50" "L.w=9" "wgrad=48" "L.w=50,L.n=15" "s:t.ffn_up:fwd:50"
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(workload, variant, steps, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "6", "--no-cpu-baseline",
           "--no-h2d-leg", "--sustained-s", "0", "--profile-steps", "0"] + extra
    cls = [v for v in variant.split(",") if v and not v.startswith(("wgrad=", "tw=", "s:"))]
    sites = ["%s:%s:-1:%s:0" % tuple(v.split(":")[1:4]) for v in variant.split(",") if v.startswith("s:")]      # s:<site>:<fwd|dgrad>:<cfg>
    if sites:
        cmd += ["--site-policy", ",".join(sites)]
    for v in variant.split(","):
        if v.startswith("wgrad="):
            cmd += ["--wgrad-cfg", v.split("=")[1]]
        if v.startswith("tw="):
            cmd += ["--wgrad-target-wgs", v.split("=")[1], "--wgrad-target-rows", "100000"]
    if cls:
        cmd += ["--class-policy", ",".join(cls)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        return None, out.stderr[-400:]
    d = json.loads(line[-1])
    return d["ms_per_step"], d["config"]["final_loss"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="plotqa-real")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("variants", nargs="*")
    a, extra = ap.parse_known_args()
    for v in ["base"] + a.variants + ["base"]:
        ms, info = run(a.workload, "" if v == "base" else v, a.steps, extra)
        print("%-40s %s  (%s)" % (v, "%.3f ms" % ms if ms else "FAILED", info), flush=True)


if __name__ == "__main__":
    main()
