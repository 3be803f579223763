"""Stand-alone timings of the 9 920-row GEMMs of the reference's PlotQA shape (bench.py --workload plotqa-real), hot, one stream: forward FFN-up with
its GELU epilogue, the FFN-down data gradient with its GELU' epilogue (and without it), FFN-down forward, FFN-up data gradient, per kernel
configuration.  us per launch and fraction of the 2.5 PFLOP/s bf16 peak."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
from crct import ops          # noqa: E402

DEV = "cuda"


def bf(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 0.05).to(DEV).bfloat16()


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    M, H, I = 9920, 768, 3072
    x, w_up, w_dn = bf(M, H, seed=1), bf(I, H, seed=2), bf(H, I, seed=3)
    b_up, b_dn = torch.zeros(I, device=DEV), torch.zeros(H, device=DEV)
    u, h = torch.empty(M, I, device=DEV, dtype=torch.bfloat16), torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
    y, dl = torch.empty(M, H, device=DEV, dtype=torch.bfloat16), bf(M, H, seed=4)
    du, dx = torch.empty(M, I, device=DEV, dtype=torch.bfloat16), torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
    ops.gemm(x, w_up, M, I, H, bias=b_up, act="gelu", preact_out=u, out=h)
    cases = [
        ("ffn_up fwd   [9920x3072x768]  bias+gelu+preact", lambda t: ops.gemm(x, w_up, M, I, H, bias=b_up, act="gelu", preact_out=u, out=h, tile=t), 2.0 * M * I * H),
        ("ffn_up fwd   [9920x3072x768]  plain", lambda t: ops.gemm(x, w_up, M, I, H, out=h, tile=t), 2.0 * M * I * H),
        ("ffn_dn dgrad [9920x3072x768]  tb, x gelu'(u)", lambda t: ops.gemm(dl, w_dn, M, I, H, tb=True, dact_src=u, dact="gelu", ld_aux=I, out=du, tile=t), 2.0 * M * I * H),
        ("ffn_dn dgrad [9920x3072x768]  tb, plain", lambda t: ops.gemm(dl, w_dn, M, I, H, tb=True, out=du, tile=t), 2.0 * M * I * H),
        ("ffn_dn fwd   [9920x768x3072]  bias+residual", lambda t: ops.gemm(h, w_dn, M, H, I, bias=b_dn, addend=x, out=y, tile=t), 2.0 * M * I * H),
        ("ffn_up dgrad [9920x768x3072]  tb, +addend", lambda t: ops.gemm(du, w_up, M, H, I, tb=True, addend=dl, out=dx, tile=t), 2.0 * M * I * H),
    ]
    dy_up, dy_dn = bf(M, I, seed=7), bf(M, H, seed=8)
    gw_up, gw_dn = torch.zeros(I, H, device=DEV), torch.zeros(H, I, device=DEV)
    cases += [
        ("ffn_up wgrad [3072x768x9920]  ta tb, fp32 out", lambda t: ops.gemm(dy_up, x, I, H, M, ta=True, tb=True, out=gw_up, out_f32=True, tile=t), 2.0 * M * I * H),
        ("ffn_dn wgrad [768x3072x9920]  ta tb, fp32 out", lambda t: ops.gemm(dy_dn, h, H, I, M, ta=True, tb=True, out=gw_dn, out_f32=True, tile=t), 2.0 * M * I * H),
    ]
    tiles = [int(t) for t in sys.argv[1:]] or [15, 9, 4, 50, 55, 48]
    print("%-52s" % "us per launch (fraction of 2.5 PF)" + "".join("%16s" % ("cfg %d" % t) for t in tiles))
    for name, fn, fl in cases:
        row = []
        for t in tiles:
            try:
                us = timeit(lambda: fn(t))
                row.append("%8.1f (%.3f)" % (us, fl / (us * 1e-6) / 2.5e15))
            except RuntimeError:
                row.append("%16s" % "-")
        print("%-52s" % name + "".join("%16s" % r for r in row), flush=True)


if __name__ == "__main__":
    main()
