"""Developer probe: does it help a LayerNorm -> FFN-up -> FFN-down chain when every activation row is produced
and consumed on ONE XCD?  crct_lab_xcd_band(256) gives XCD x the row tiles of band x (and every column tile) in the GEMM tile maps and
the rows of band x in the LayerNorm forward; 0 = the product's rectangle maps / round-robin rows.  M = 2048 (8 bands of two 128-row
tiles: the clean case; at M = 1600 a band is 200 rows = 1.56 tiles), H = 768, I = 3072, cold weights (48 rotating copies) behind a
256 MB copy that flushes L2 / Infinity Cache, like tools/coldstart_lab.py.

    python tools/lab/band_lab.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch

from crct import lib as L, ops

lib = L.load()
dev = "cuda"
torch.manual_seed(0)
H, I = 768, 3072
NCOPY = 24


def bf(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(torch.bfloat16)


big, big2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev), torch.empty(256 << 20, dtype=torch.uint8, device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
w1 = [bf(I, H, scale=0.05) for _ in range(NCOPY)]
w2 = [bf(H, I, scale=0.02) for _ in range(NCOPY)]


def read(tile, kind=0):
    cnt, fl, ms = C.c_long(), C.c_double(), C.c_double()
    lib.crct_prof_read(tile * 3 + kind, C.byref(cnt), C.byref(fl), C.byref(ms))
    return ms.value * 1e3, cnt.value


XS = {}


def chain(M, band, tile_up, tile_dn, flush, iters=30):
    x = XS.setdefault(M, bf(M, H))
    h = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    s = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    lib.crct_lab_xcd_band(band)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def once(i, prof):
        if flush:
            big2.copy_(big)
        e0.record()
        a = ops.layernorm_fwd(x, gamma, beta)[0]
        lib.crct_prof_enable(1 if prof else 0)
        ops.gemm(a, w1[i % NCOPY], M, I, H, tile=tile_up, out=h, act="gelu")
        ops.gemm(h, w2[i % NCOPY], M, H, I, tile=tile_dn, out=s, addend=x)
        lib.crct_prof_enable(0)
        y = ops.layernorm_fwd(s, gamma, beta)[0]
        e1.record()
        return y
    for i in range(4):
        y = once(i, False)
    torch.cuda.synchronize()
    lib.crct_prof_reset()
    tot = 0.0
    for i in range(iters):
        y = once(i, True)
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    up, n1 = read(tile_up)
    if tile_dn == tile_up:
        per = up / max(n1, 1)
        res = "up+down %6.1f us per GEMM pair" % (2 * per)
    else:
        dn, n2 = read(tile_dn)
        res = "up %6.1f us  down %6.1f us" % (up / max(n1, 1), dn / max(n2, 1))
    lib.crct_prof_reset()
    lib.crct_lab_xcd_band(0)
    return tot / iters, res, y.float().abs().sum().item()


for M in (2048, 1600):
    for flush in (True, False):
        for tile_up, tile_dn in ((12, 15), (46, 46), (4, 15), (48, 46)):
            rows = []
            for band in (0, 256):
                t, res, chk = chain(M, band, tile_up, tile_dn, flush)
                rows.append((band, t, res, chk))
            same = rows[0][3] == rows[1][3]
            for band, t, res, chk in rows:
                print("M %4d  %-5s  cfg up %2d / down %2d  band %3d : LN+up+down+LN %6.1f us  (%s)  %s" %
                      (M, "cold" if flush else "hot", tile_up, tile_dn, band, t, res, "" if same else "CHECKSUM DIFFERS"))
