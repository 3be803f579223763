"""What the step gains when a class of kernels costs NOTHING: the training step (configs[1]) on the LAB build of the library
(make -C cqa-crct_amd/csrc lab -> tools/lab/libcrct_lab.so) with the launches of one class left out (CRCT_LAB_SKIP; wrong results, timing
only), against the same build with nothing left out.  gain / kernel time of the class (from rocprof's per-kernel totals of the round's
collection) says how much of a class is on the step's critical path -- the map of where a faster kernel would pay.
    python tools/lab/step_sensitivity.py [--reps 2]"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "tools", "lab", "libcrct_lab.so")


SHAPE = []      # extra step_time.py arguments (--batch / --vis / --tokens): set from --shape


def run(skip, reps, report=False, extra=None):
    env = dict(os.environ)
    env.pop("CRCT_LAB_SKIP", None)
    env.update(extra or {})
    if skip:
        env["CRCT_LAB_SKIP"] = skip
    if report:
        env["CRCT_LAB_REPORT"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "lab", "step_time.py"), "--lib", LIB, "--reps", str(reps)] + SHAPE + (["--steps", "4"] if report else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    m = re.search(r"step_time.*: ([0-9., ]+) ms", p.stdout)
    times = [float(x) for x in m.group(1).split(",")] if m else []
    return times, p.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--set", default="classes", help="classes: one kernel class left out at a time; interference: what KIND of load the side work is")
    ap.add_argument("--shape", default="", help="B,V,T of the step (default: configs[1] = 80,36,20); 80,44,124 = the reference's PlotQA shape")
    a = ap.parse_args()
    if a.shape:
        B, V, T = a.shape.split(",")
        SHAPE.extend(["--batch", B, "--vis", V, "--tokens", T])
    _, err = run("", 1, report=True)
    streams = {}
    for line in err.splitlines():
        m = re.match(r"\[crct lab\] stream (\d+) (\S+)\s+launched (\d+)", line)
        if m:
            streams.setdefault(int(m.group(1)), {})[m.group(2)] = int(m.group(3))
    role = {}
    for k, names in streams.items():
        j = " ".join(names)
        if "embed_text_fwd" in j: role["text"] = k
        elif "embed_image_fwd" in j: role["visual"] = k
        elif "adamw_kernel" in j: role["aux"] = k
    groups = sorted((sum(v for n, v in names.items() if "gemm_group" in n), k) for k, names in streams.items() if k not in role.values())
    for (cnt, k), nm in zip(reversed(groups), ("wgrad_text", "wgrad_visual")):
        role[nm] = k
    print("streams by first use:", role, flush=True)
    t, v, wt, wv = role["text"], role["visual"], role["wgrad_text"], role["wgrad_visual"]
    cases = [
        ("nothing left out", ""),
        ("AdamW", "adamw_kernel"),
        ("grouped weight gradients (both streams)", "gemm_group"),
        ("everything on the weight-gradient streams", "*@%d,*@%d" % (wt, wv)),
        ("weight-gradient streams + AdamW", "*@%d,*@%d,adamw_kernel" % (wt, wv)),
        ("column-sum finalize passes", "finalize_partials"),
        ("LayerNorm backward (both streams)", "ln_bwd_kernel"),
        ("LayerNorm backward, text", "ln_bwd_kernel@%d" % t),
        ("LayerNorm forward (both)", "ln_fwd_kernel"),
        ("attention backward (both)", "attn_bwd"),
        ("attention forward (both)", "attn_fwd"),
        ("every GEMM of the text stream", "gemm_pipe@%d,gemm_kernel@%d,gemm_ldr@%d" % (t, t, t)),
        ("every GEMM of the visual stream", "gemm_pipe@%d,gemm_kernel@%d,gemm_ldr@%d" % (v, v, v)),
        ("everything on the visual stream", "*@%d" % v),
        ("everything on the text stream", "*@%d" % t),
        ("everything but the text stream", "*@%d,*@%d,*@%d,adamw_kernel" % (v, wt, wv)),
        ("everything but the visual stream", "*@%d,*@%d,*@%d,adamw_kernel" % (t, wt, wv)),
        ("nothing left out (again)", ""),
    ]
    if a.set == "interference":
        cases = [
            ("nothing left out", ""),
            ("AdamW left out", "adamw_kernel"),
            ("AdamW as a memory-free spin (paced as 4.5 TB/s)", "", {"CRCT_LAB_ADAMW_SPIN": "4.5"}),
            ("AdamW as a memory-free spin (paced as 3 TB/s)", "", {"CRCT_LAB_ADAMW_SPIN": "3.0"}),
            ("grouped weight gradients left out", "gemm_group"),
            ("grouped weight gradients: no operand DMA", "", {"CRCT_GEMM_DBG_GROUP": "2"}),
            ("grouped weight gradients: no epilogue (no stores)", "", {"CRCT_GEMM_DBG_GROUP": "1"}),
            ("grouped weight gradients: neither", "", {"CRCT_GEMM_DBG_GROUP": "3"}),
            ("every GEMM: no operand DMA", "", {"CRCT_GEMM_DBG": "2"}),
            ("every GEMM: no epilogue", "", {"CRCT_GEMM_DBG": "1"}),
            ("nothing left out (again)", ""),
        ]
    if a.set == "adamw":
        cases = [
            ("nothing left out", ""),
            ("AdamW left out", "adamw_kernel"),
            ("the forward does not wait for AdamW", "", {"CRCT_LAB_NO_PARAM_WAIT": "1"}),
            ("AdamW as a sleeping kernel (paced as 4.5 TB/s)", "", {"CRCT_LAB_ADAMW_SPIN": "4.5"}),
            ("... of 64 threads per workgroup", "", {"CRCT_LAB_ADAMW_SPIN": "4.5", "CRCT_LAB_ADAMW_SPIN_THREADS": "64"}),
            ("... and the forward does not wait for it", "", {"CRCT_LAB_ADAMW_SPIN": "4.5", "CRCT_LAB_NO_PARAM_WAIT": "1"}),
            ("AdamW as a sleeping kernel (paced as 9 TB/s)", "", {"CRCT_LAB_ADAMW_SPIN": "9.0"}),
            ("AdamW as a sleeping kernel (paced as 3 TB/s)", "", {"CRCT_LAB_ADAMW_SPIN": "3.0"}),
            ("nothing left out (again)", ""),
        ]
    if a.set == "spin":
        def sp(g, t):
            return {"CRCT_LAB_ADAMW_SPIN": "4.5", "CRCT_LAB_ADAMW_SPIN_GRID": str(g), "CRCT_LAB_ADAMW_SPIN_THREADS": str(t)}
        cases = [("nothing left out", ""), ("AdamW left out", "adamw_kernel")]
        for g, t in ((256, 256), (256, 64), (256, 128), (64, 256), (128, 256), (1024, 64), (512, 64), (128, 512), (64, 1024), (32, 1024), (512, 256)):
            cases.append(("sleeping stand-in: %d workgroups x %d threads" % (g, t), "", sp(g, t)))
        cases.append(("nothing left out (again)", ""))
    if a.set == "small":
        cases = [
            ("nothing left out", ""),
            ("the head chain's 64 x 64-tile GEMMs (forward, data gradients)", "gemm_pipe_kernelILi2ELi2ELi2ELi2E"),
            ("head chain GEMMs + weight gradients + bias sums", "gemm_pipe_kernelILi2ELi2ELi2ELi2E,gemm_kernelILi2ELi2E,colsum_kernel"),
            ("head row kernels (loss, pooling)", "head_"),
            ("embedding kernels (forward + backward)", "embed_"),
            ("image-feature softmax + gather", "softmax_rows,gather_sum"),
            ("128 x 128-tile GEMMs of the data streams (visual QKV, image embedding)", "gemm_pipe_kernelILi4ELi4ELi2ELi4E"),
            ("nothing left out (again)", ""),
        ]
    base = None
    for case in cases:
        name, skip = case[0], case[1]
        extra = case[2] if len(case) > 2 else None
        times, err = run(skip, a.reps, extra=extra)
        if not times:
            print("%-48s FAILED: %s" % (name, err[-300:]), flush=True)
            continue
        best = min(times)
        if base is None:
            base = best
        print("%-52s %s ms  (gain %.3f)   CRCT_LAB_SKIP=%s %s" % (name, ", ".join("%.3f" % x for x in times), base - best, skip, extra or ""), flush=True)


if __name__ == "__main__":
    main()
