"""VERDICT r5 item 3, lab: does giving every stream its own XCDs help the three kinds of work that share the chip in the step?

Three HIP streams, each repeating its own launch sequence at configs[1]'s shapes (B 80: 1600 text rows, 2880 visual rows):
  text    FFN-up (1600 x 3072 x 768, GELU + saved pre-activation) -> FFN-down (1600 x 768 x 3072, residual add)      [the dependent chain]
  visual  FFN-up (2880 x 1024 x 1024, GELU) -> FFN-down (2880 x 1024 x 1024, residual add)
  wgrad   a text layer's four weight gradients as ONE grouped launch (contraction over the 1600 rows)
run (a) alone, (b) all three on the shared chip, (c) all three with CrctGemmArgs.xcd_mask giving each stream a disjoint set of XCDs.
Reported: per stream the time for its R repetitions (us per repetition) and the wall time of the trio.  `--pmc` prints nothing extra:
run the script under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace` to get per-kernel L2 hit rates for each mode (one mode
per process: --only shared | 332 | 422 | 431 | alone)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
import ctypes as C             # noqa: E402
from crct import ops, lib as L # noqa: E402

LIB = L.load()

DEV = "cuda"


def bf(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 0.05).to(DEV).bfloat16()


class Text(object):
    """Argument structs are built ONCE per mask; a repetition is two ctypes calls (the host must not be the bottleneck of the lab)."""

    def __init__(self, M=1600, H=768, I=3072):
        self.M, self.H, self.I = M, H, I
        self.x, self.w1, self.w2 = bf(M, H, seed=1), bf(I, H, seed=2), bf(H, I, seed=3)
        self.b1, self.b2 = torch.zeros(I, device=DEV), torch.zeros(H, device=DEV)
        self.u = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
        self.h = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
        self.y = torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
        self.cache = {}

    def args(self, mask):
        if mask not in self.cache:
            g1, g2 = L.GemmArgs(), L.GemmArgs()
            ops._gemm_args(g1, self.x, self.w1, self.M, self.I, self.H, bias=self.b1, act="gelu", preact_out=self.u, out=self.h, xcd_mask=mask)
            ops._gemm_args(g2, self.h, self.w2, self.M, self.H, self.I, bias=self.b2, addend=self.x, out=self.y, xcd_mask=mask)
            self.cache[mask] = (g1, g2)
        return self.cache[mask]

    def run(self, mask, stream):
        g1, g2 = self.args(mask)
        LIB.crct_gemm_bf16(C.byref(g1), stream)
        LIB.crct_gemm_bf16(C.byref(g2), stream)


class Visual(Text):
    def __init__(self):
        Text.__init__(self, 2880, 1024, 1024)


class Wgrad(object):
    """dW = dy^T x of FFN-up, FFN-down, attention output and QKV of a text layer: one grouped launch."""

    def __init__(self, R=1600, H=768, I=3072):
        self.R = R
        self.probs = []
        for n, k, seed in ((I, H, 11), (H, I, 12), (H, H, 13), (3 * H, H, 14)):
            dy, x = bf(R, n, seed=seed), bf(R, k, seed=seed + 50)
            self.probs.append(dict(A=dy, B=x, M=n, N=k, K=R, ta=True, tb=True, out=torch.zeros(n, k, device=DEV), out_f32=True))
        self.cache = {}

    def run(self, mask, stream):
        if mask not in self.cache:
            arr = (L.GemmArgs * len(self.probs))()
            for g, p in zip(arr, self.probs):
                ops._gemm_args(g, **dict(p, xcd_mask=mask))
            self.cache[mask] = arr
        LIB.crct_gemm_bf16_grouped(self.cache[mask], len(self.probs), stream)


def timed(jobs, masks, reps):
    """jobs: [(name, obj)], masks: [int] -> ({name: us per repetition}, wall us per repetition, host enqueue us per repetition)"""
    streams = [torch.cuda.Stream() for _ in jobs]
    ptrs = [C.c_void_p(st.cuda_stream) for st in streams]
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in jobs]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in jobs]
    for (_, o), m, sp in zip(jobs, masks, ptrs):          # warm-up (first launches set kernel attributes)
        o.run(m, sp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(len(jobs)):
        ev0[i].record(streams[i])
    for _ in range(reps):                       # interleaved enqueue, like the engine's host thread
        for (_, o), m, sp in zip(jobs, masks, ptrs):
            o.run(m, sp)
    host = (time.perf_counter() - t0) * 1e6 / reps
    for i in range(len(jobs)):
        ev1[i].record(streams[i])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6 / reps
    return {n: ev0[i].elapsed_time(ev1[i]) * 1e3 / reps for i, (n, _) in enumerate(jobs)}, wall, host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    jobs = [("text", Text()), ("visual", Visual()), ("wgrad", Wgrad())]
    modes = [("shared", [0, 0, 0]),
             ("332  text 0-2 | visual 3-5 | wgrad 6-7", [0b00000111, 0b00111000, 0b11000000]),
             ("422  text 0-3 | visual 4-5 | wgrad 6-7", [0b00001111, 0b00110000, 0b11000000]),
             ("431  text 0-3 | visual 4-6 | wgrad 7", [0b00001111, 0b01110000, 0b10000000]),
             ("text 0-4 | visual+wgrad shared 5-7", [0b00011111, 0b11100000, 0b11100000]),
             ("text all | visual+wgrad 4-7", [0, 0b11110000, 0b11110000])]
    if not a.only or a.only == "alone":
        for j in jobs:
            per, wall, host = timed([j], [0], a.reps)
            print("alone   %-7s %8.1f us per repetition (host enqueue %.1f)" % (j[0], per[j[0]], host), flush=True)
        for nm, m in (("text on 4 XCDs", 0b1111), ("text on 3 XCDs", 0b111)):
            per, wall, host = timed([jobs[0]], [m], a.reps)
            print("alone   %-14s %8.1f us per repetition" % (nm, per["text"]), flush=True)
    for name, masks in modes:
        if a.only and a.only != name.split()[0]:
            continue
        for rep in range(2):
            per, wall, host = timed(jobs, masks, a.reps)
            print("%-46s text %7.1f  visual %7.1f  wgrad %7.1f  | trio wall %7.1f us per repetition (host enqueue %.1f)" % (name, per["text"], per["visual"], per["wgrad"], wall, host),
                  flush=True)


if __name__ == "__main__":
    main()
