"""Lab: the residual GEMMs (attention-output / FFN-down projections: y = dropout(x W^T + b) + r) stand-alone by graph replay, with the bf16
residual / bf16 sum of rounds 1 - 5 against the fp32 residual / fp32 sum of the fp32 residual stream (CrctGemmArgs.addend_f32 / c_cached).
    python tools/lab/residual_gemm_standalone.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cqa-crct_amd"))
from crct import ops, lib as L  # noqa: E402

DEV = torch.device("cuda", 0)


def graph_time(fn, n=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        gr.replay()
        gr.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * n) * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    lib = L.load()
    for name, M, N, K in (("t.out", 1600, 768, 768), ("t.ffn_down", 1600, 768, 3072), ("v.out / v.ffn_down", 2880, 1024, 1024), ("c.out_t", 1600, 768, 1024),
                          ("t.ffn_down, PlotQA shape", 9920, 768, 3072)):
        x = torch.randn(M, K, generator=g).to(DEV).bfloat16()
        w = (torch.randn(N, K, generator=g) * 0.05).to(DEV).bfloat16()
        b = torch.randn(N, generator=g).to(DEV)
        r32 = torch.randn(M, N, generator=g).to(DEV)
        r16 = r32.bfloat16()
        o16, o32 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, N, device=DEV)
        res = []
        for addend, out, cached in ((r16, o16, False), (r32, o32, True), (r32, o32, False), (None, o16, False)):
            ga = L.GemmArgs()
            ops._gemm_args(ga, x, w, M, N, K, out=out, bias=b, addend=addend, p_drop=0.1, site=3, seed=5, c_cached=cached)
            res.append(graph_time(lambda: L.check(lib.crct_gemm_bf16(C.byref(ga), L.current_stream()))))
        print("%-26s %5d x %4d x %4d   bf16 residual + bf16 sum %6.1f us   fp32 residual + fp32 sum %6.1f us (streaming stores %6.1f)   no residual, bf16 out %6.1f us"
              % (name, M, N, K, res[0], res[1], res[2], res[3]))


if __name__ == "__main__":
    main()
