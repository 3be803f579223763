"""Developer probe: why does a GEMM of the step take ~5 us longer IN the step than back to back with itself?
Times ONE GEMM (kernel begin / end stamps, crct_prof_*) after different predecessors on the same stream.

    python tools/coldstart_lab.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch

from crct import lib as L, ops

lib = L.load()
dev = "cuda"
torch.manual_seed(0)
M, N, K = 1600, 768, 768
NCOPY = 48


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


xs = [bf(M, K) for _ in range(NCOPY)]
ws = [bf(N, K, scale=0.05) for _ in range(NCOPY)]
x2, w2 = bf(M, 768), bf(3072, 768, scale=0.05)
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
ln_x = bf(M, 768)
gamma, beta = torch.ones(768, device=dev), torch.zeros(768, device=dev)
q = bf(80, 20, 3 * 768)
qq, kk, vv = q[..., :768].contiguous(), q[..., 768:1536].contiguous(), q[..., 1536:].contiguous()
km = torch.ones(80, 20, dtype=torch.uint8, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)


def measure(name, pre, fresh=False, tile=12, iters=40):
    def gemm(i):
        a, b = (xs[i % NCOPY], ws[i % NCOPY]) if fresh else (xs[0], ws[0])
        ops.gemm(a, b, M, N, K, tile=tile, out=out)
    for i in range(5):
        pre(i)
        gemm(i)
    torch.cuda.synchronize()
    lib.crct_prof_reset()
    for i in range(iters):
        pre(i)
        lib.crct_prof_enable(1)
        gemm(i)
        lib.crct_prof_enable(0)
    torch.cuda.synchronize()
    cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
    lib.crct_prof_read(tile * 3, C.byref(cnt), C.byref(fl), C.byref(ms))
    print("%-64s %6.2f us per launch (%d launches)" % (name, ms.value * 1e3 / max(cnt.value, 1), cnt.value))
    lib.crct_prof_reset()


nothing = lambda i: None
measure("GEMM t.out fwd back to back, same operands", nothing)
measure("... fresh operands every launch (48 copies)", nothing, fresh=True)
measure("after a LayerNorm forward", lambda i: ops.layernorm_fwd(ln_x, gamma, beta))
measure("after an attention forward", lambda i: ops.attention_fwd(qq, kk, vv, km, 12, 64))
measure("after another GEMM of the same kernel (t.ffn_up shape)", lambda i: ops.gemm(x2, w2, M, 3072, 768, tile=12))
measure("after a GEMM of another kernel (tile 15)", lambda i: ops.gemm(x2, w2, M, 3072, 768, tile=15))
measure("after a 256 MB device copy (L2 / Infinity Cache flushed)", lambda i: big2.copy_(big))
measure("after a 256 MB copy, fresh operands", lambda i: big2.copy_(big), fresh=True)
measure("after LN + attention + other GEMM, fresh operands",
        lambda i: (ops.layernorm_fwd(ln_x, gamma, beta), ops.attention_fwd(qq, kk, vv, km, 12, 64),
                   ops.gemm(x2, w2, M, 3072, 768, tile=15)), fresh=True)
side = torch.cuda.Stream()


def hog(i):
    with torch.cuda.stream(side):
        big2.copy_(big)


measure("beside a 256 MB copy on another stream (HBM busy)", hog)
torch.cuda.synchronize()
measure("GEMM back to back again", nothing)

# ---- which kernel configuration copes best with operands that come from HBM (the state of the step)?
print("\nafter a 256 MB copy (cold operands), per configuration: us per launch")
shapes = [("t.out fwd", 1600, 768, 768, False), ("t.ffn_dn fwd", 1600, 768, 3072, False), ("t.ffn_up fwd", 1600, 3072, 768, False),
          ("t.qkv fwd", 1600, 2304, 768, False), ("t.ffn_up dg", 1600, 768, 3072, True), ("v.ffn fwd", 2880, 1024, 1024, False)]
tiles = [12, 15, 13, 14, 10, 4, 9, 3]
print("%-14s" % "shape" + "".join("%8d" % t for t in tiles) + "   | hot: " + "".join("%8d" % t for t in tiles[:4]))
for name, m, n, k, tb in shapes:
    a = bf(m, k)
    b = bf(k, n, scale=0.05) if tb else bf(n, k, scale=0.05)
    o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    row = []
    for cold in (True, False):
        for t in (tiles if cold else tiles[:4]):
            for i in range(3):
                ops.gemm(a, b, m, n, k, tb=tb, tile=t, out=o)
            torch.cuda.synchronize()
            lib.crct_prof_reset()
            for i in range(20):
                if cold:
                    big2.copy_(big)
                lib.crct_prof_enable(1)
                ops.gemm(a, b, m, n, k, tb=tb, tile=t, out=o)
                lib.crct_prof_enable(0)
            torch.cuda.synchronize()
            tot = 0.0
            for kind in range(3):
                cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
                lib.crct_prof_read(t * 3 + kind, C.byref(cnt), C.byref(fl), C.byref(ms))
                tot += ms.value * 1e3 / max(cnt.value, 1) if cnt.value else 0.0
            row.append(tot)
    print("%-14s" % name + "".join("%8.1f" % v for v in row[:len(tiles)]) + "   |      " + "".join("%8.1f" % v for v in row[len(tiles):]))

# ---- does touching the WEIGHTS ahead of time (crct_prefetch) take the cold start away?
print("\nt.out fwd / t.ffn_dn fwd after a 256 MB copy, then a prefetch of ... (us per launch, cfg 12 / 15)")
for name, m, n, k, t in (("t.out fwd", 1600, 768, 768, 12), ("t.ffn_dn fwd", 1600, 768, 3072, 15), ("t.ffn_up fwd", 1600, 3072, 768, 12)):
    a, b = bf(m, k), bf(n, k, scale=0.05)
    o = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for what in ("nothing", "weights", "activations", "both", "weights on another stream, 100 us earlier"):
        lib.crct_prof_reset()
        for i in range(23):
            big2.copy_(big)
            if what in ("weights", "both"):
                lib.crct_prefetch(b.data_ptr(), b.numel() * 2, 32, L.current_stream())
            if what in ("activations", "both"):
                lib.crct_prefetch(a.data_ptr(), a.numel() * 2, 32, L.current_stream())
            if what.startswith("weights on"):
                side.wait_stream(torch.cuda.current_stream())
                lib.crct_prefetch(b.data_ptr(), b.numel() * 2, 16, side.cuda_stream)
                ops.gemm(x2, w2, M, 3072, 768, tile=12)           # ~15-20 us of other work on the main stream meanwhile
                ops.gemm(x2, w2, M, 3072, 768, tile=12)
                ops.gemm(x2, w2, M, 3072, 768, tile=12)
            if i >= 3:
                lib.crct_prof_enable(1)
            ops.gemm(a, b, m, n, k, tile=t, out=o)
            lib.crct_prof_enable(0)
        torch.cuda.synchronize()
        cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
        lib.crct_prof_read(t * 3, C.byref(cnt), C.byref(fl), C.byref(ms))
        print("  %-14s prefetch %-44s %6.2f us (%d launches)" % (name, what, ms.value * 1e3 / max(cnt.value, 1), cnt.value))
