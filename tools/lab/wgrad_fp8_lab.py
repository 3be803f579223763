"""Grouped weight gradients of one text / visual layer: bf16 kernel vs the fp8 (token-major, transposing LDS read) kernel,
hot caches and with a 512 MB flush between launches.  Developer tooling."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cqa-crct_amd"))
from crct import ops, lib as L

DEV = torch.device("cuda:0")


def layer(R, shapes):
    g = torch.Generator(device="cpu").manual_seed(1)
    b16, f8 = [], []
    for N, K in shapes:
        dy, x = torch.randn(R, N, generator=g) * 2e-3, torch.randn(R, K, generator=g)
        s_dy, s_x = 57344.0 / float(dy.abs().max()), 448.0 / float(x.abs().max())
        out = torch.zeros(N, K, device=DEV)
        b16.append((dy.to(torch.bfloat16).to(DEV), x.to(torch.bfloat16).to(DEV), out))
        f8.append(((dy * s_dy).to(torch.float8_e5m2).to(DEV), (x * s_x).to(torch.float8_e4m3fn).to(DEV),
                   torch.tensor([s_dy], device=DEV), torch.tensor([s_x], device=DEV), torch.zeros(N, K, device=DEV)))
    return b16, f8


def timed(fn, flush=None, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3


flush = torch.zeros(128 * 1024 * 1024, device=DEV)
for name, R, shapes in [("text layer", 1600, [(3072, 768), (768, 3072), (768, 768), (2304, 768)]),
                        ("visual layer", 2880, [(1024, 1024), (1024, 1024), (1024, 1024), (3072, 1024)]),
                        ("text FFN pair", 1600, [(3072, 768), (768, 3072)])]:
    b16, f8 = layer(R, shapes)
    gf = sum(2.0 * R * n * k for n, k in shapes) / 1e9
    for label, fl in (("hot", None), ("flushed", flush)):
        t0 = timed(lambda: ops.gemm_wgrad_grouped(b16), fl)
        t1 = timed(lambda: ops.gemm_wgrad_fp8(f8, accumulate=True, tile=36), fl)
        t2 = timed(lambda: ops.gemm_wgrad_fp8(f8, accumulate=True, tile=37), fl)
        t3 = timed(lambda: ops.gemm_wgrad_fp8(f8, accumulate=False, tile=36), fl)
        print("%-14s %-8s %6.2f GF   bf16 %6.1f us (%5.0f TF)   fp8 s3 %6.1f us (%5.0f TF)   fp8 s2 %6.1f us   fp8 s3 overwrite %6.1f us"
              % (name, label, gf, t0, gf / t0 * 1e3, t1, gf / t1 * 1e3, t2, t3))
