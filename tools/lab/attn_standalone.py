"""Lab: the attention kernels STAND-ALONE at the step's shapes (hot, one stream, dropout 0.1), beside what they take inside the step
(profiles/r6_*kernel_stats*.csv); the backward with and without the forward's row statistics.  The short shapes are bounded by this
wrapper's host cost (~11 us per call: output allocations), the long ones by the GPU.
    python tools/lab/attn_standalone.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
from crct import ops  # noqa: E402

DEV = torch.device("cuda", 0)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    g = torch.Generator(device="cpu").manual_seed(0)
    print("# us per launch, stand-alone (50 back-to-back launches on one stream; allocations of the wrapper included)")
    for B, heads, Tq, Tk, d in ((80, 16, 20, 20, 48), (80, 16, 36, 36, 64), (80, 32, 20, 36, 32), (80, 16, 124, 124, 48), (80, 32, 124, 44, 32),
                                (80, 32, 44, 124, 32), (80, 16, 44, 44, 64)):
        Hh = heads * d
        q = torch.randn(B, Tq, Hh, generator=g).to(DEV).bfloat16()
        k = torch.randn(B, Tk, Hh, generator=g).to(DEV).bfloat16()
        v = torch.randn(B, Tk, Hh, generator=g).to(DEV).bfloat16()
        dctx = torch.randn(B, Tq, Hh, generator=g).to(DEV).bfloat16()
        km = torch.ones(B, Tk, dtype=torch.uint8, device=DEV)
        lse = torch.empty(B, heads, Tq, device=DEV)
        ctx = ops.attention_fwd(q, k, v, km, heads, d, p_drop=0.1, site=1, seed=3, row_lse=lse)
        t_f = timed(lambda: ops.attention_fwd(q, k, v, km, heads, d, p_drop=0.1, site=1, seed=3, row_lse=lse))
        t_b = timed(lambda: ops.attention_bwd(q, k, v, km, dctx, heads, d, p_drop=0.1, site=1, seed=3))
        t_bk = timed(lambda: ops.attention_bwd(q, k, v, km, dctx, heads, d, p_drop=0.1, site=1, seed=3, row_lse=lse, ctx=ctx))
        gf = 4.0 * B * heads * Tq * Tk * d / 1e9
        print("attention B %d heads %2d %3d x %3d d %2d   fwd %6.1f us (%5.1f TFLOP/s)   bwd %6.1f us, with kept statistics %6.1f us (%5.1f TFLOP/s)"
              % (B, heads, Tq, Tk, d, t_f, gf / t_f * 1e3, t_b, t_bk, 2.5 * gf / t_bk * 1e3))


if __name__ == "__main__":
    main()
