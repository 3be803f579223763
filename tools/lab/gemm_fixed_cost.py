"""Fixed cost of a GEMM launch: kernel durations (begin / end stamps of the library's own profiler, no host time in them) of the bf16 and
fp8 forward kernels at the CRCT output shapes for contraction lengths 128 ... 3072, hot; a straight-line fit gives the cost at K = 0
(dispatch, descriptors, first DMA round trip, epilogue) and the cost per 128 of K.  Developer tooling."""
import ctypes as C
import os
import sys

import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cqa-crct_amd"))
from crct import ops, lib as L

DEV = torch.device("cuda:0")
lib = L.load()


def kernel_us(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.crct_prof_reset()
    lib.crct_prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    lib.crct_prof_enable(0)
    tot, cnt_all = 0.0, 0
    for v in range(120):
        cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
        if lib.crct_prof_read(v, C.byref(cnt), C.byref(fl), C.byref(ms)) == 0 and cnt.value > 0:
            tot += ms.value * 1e3
            cnt_all += cnt.value
    lib.crct_prof_reset()
    return tot / max(cnt_all, 1)


for M, N in ((1600, 768), (1600, 3072), (2880, 1024), (2880, 3072)):
    rows = []
    for K in (128, 256, 512, 768, 1536, 3072):
        a = torch.randn(M, K, device=DEV).bfloat16()
        b = torch.randn(N, K, device=DEV).bfloat16()
        a8, b8 = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn)
        one = torch.ones(1, device=DEV)
        t16 = kernel_us(lambda: ops.gemm(a, b, M, N, K))
        t8 = kernel_us(lambda: ops.gemm_fp8(a8, b8, one, one, M, N, K))
        rows.append((K, t16, t8))
    # least squares t = c0 + c1 * (K / 128)
    def fit(idx):
        xs = [r[0] / 128.0 for r in rows]; ys = [r[idx] for r in rows]
        n = len(xs); mx, my = sum(xs) / n, sum(ys) / n
        c1 = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        return my - c1 * mx, c1
    f16, f8 = fit(1), fit(2)
    print("M=%d N=%d  " % (M, N) + "  ".join("K=%d: %.1f / %.1f us" % r for r in rows))
    print("          bf16: %.1f us + %.2f us per 128 of K;   fp8: %.1f us + %.2f us per 128 of K" % (f16[0], f16[1], f8[0], f8[1]))
