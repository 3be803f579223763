"""Lab: the LayerNorm kernels STAND-ALONE at the step's row counts -- 40 launches captured in a graph and replayed (GPU time only: the
Python wrappers' ~11 us of host cost per call would hide kernels this short).  Operands are re-read every launch, so they sit in the
Infinity Cache: an upper bound on what the kernels reach in the step.
    python tools/lab/ln_standalone.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cqa-crct_amd"))
from crct import ops, lib as L
DEV = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
def graph_time(fn, n=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); gr.replay(); gr.replay(); b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * n) * 1e3
lib = L.load()
for M, H in ((1600, 768), (2880, 1024), (9920, 768), (3520, 1024)):
    x = torch.randn(M, H, generator=g).to(DEV).bfloat16(); dy = torch.randn(M, H, generator=g).to(DEV).bfloat16()
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    dx, dxl = torch.empty_like(x), torch.empty_like(x)
    nb = lib.crct_layernorm_bwd_blocks(M)
    part = torch.empty(3 * 4 * nb * H, device=DEV); dg, db, dbi = (torch.zeros(H, device=DEV) for _ in range(3))
    thr, sc, st = ops._drop(0.1, 3)
    def fwd():
        L.check(lib.crct_layernorm_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd), M, H, 1e-12, 0, 1.0, 0, 0, L.current_stream()))
    def bwd():
        L.check(lib.crct_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dx), L.ptr(dxl), L.ptr(dg), L.ptr(db), L.ptr(dbi), L.ptr(part), M, H, 0,
                                       0, 1.0, 0, thr, sc, st, 5, L.current_stream()))
    tf, tb = graph_time(fwd), graph_time(bwd)
    print("layernorm %5d x %4d  blocks %d  fwd %5.1f us (%.2f TB/s)   bwd rows+finalize %5.1f us (%.2f TB/s of 4 MH2 bytes)" % (M, H, nb, tf, 2*M*H*2/tf/1e6, tb, 4*M*H*2/tb/1e6))
