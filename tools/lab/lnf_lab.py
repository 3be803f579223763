"""Stand-alone cost of the folded LayerNorm pieces (hot operands, nothing beside them): producer GEMM with / without partial statistics,
LayerNorm kernel, consumer GEMM plain / folded, and the fold kernel.   python tools/lab/lnf_lab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch                       # noqa: E402
from crct import ops               # noqa: E402


def timed(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = "cuda"
    for name, M, H, N, Kp in (("t.out->t.ffn_up", 1600, 768, 3072, 768), ("t.ffn_down->t.qkv", 1600, 768, 2304, 3072),
                              ("v.ffn_down->v.qkv", 2880, 1024, 3072, 1024), ("v.out->v.ffn_up", 2880, 1024, 1024, 1024)):
        x0 = (torch.randn(M, Kp, device=dev) * 0.5).bfloat16()
        w0 = (torch.randn(H, Kp, device=dev) * 0.05).bfloat16()
        b0 = torch.randn(H, device=dev) * 0.1
        res = torch.randn(M, H, device=dev).bfloat16()
        stats = torch.zeros(M, 16, 2, device=dev)
        s_out = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
        t_prod = timed(lambda: ops.gemm(x0, w0, M, H, Kp, bias=b0, addend=res, out=s_out, p_drop=0.1, site=3, seed=5))
        t_prod_s = timed(lambda: ops.gemm(x0, w0, M, H, Kp, bias=b0, addend=res, out=s_out, p_drop=0.1, site=3, seed=5, ln_stats_out=stats))
        gamma, beta = 1.0 + 0.1 * torch.randn(H, device=dev), 0.1 * torch.randn(H, device=dev)
        t_ln = timed(lambda: ops.layernorm_fwd(s_out, gamma, beta))
        W = torch.randn(N, H, device=dev) * 0.05
        b = torch.randn(N, device=dev) * 0.1
        flat = torch.cat([W.flatten(), b, gamma, beta]).contiguous()
        wfold, cvec, bvec, _ = ops.ln_fold_weights(flat, [(0, N * H, N * H + N, N * H + N + H, H, N)])
        wf = wfold[:N * H].view(N, H).contiguous()
        wb = W.bfloat16()
        y, _, _ = ops.layernorm_fwd(s_out, gamma, beta)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        tiles = H // 64
        lnf = dict(stats=stats, tiles=tiles, c=cvec, gamma=gamma, beta=beta, y=y, mean=mean, rstd=rstd)
        t_plain = timed(lambda: ops.gemm(y, wb, M, N, H, bias=b, out=out, act="gelu", preact_out=pre))
        t_lnf = timed(lambda: ops.gemm(s_out, wf, M, N, H, bias=bvec, out=out, act="gelu", preact_out=pre, lnf=lnf))
        import ctypes
        from crct import lib as L
        lib = L.load()
        if hasattr(lib, "crct_lab_lnf_dbg"):
            abl = {}
            for d in (1, 2, 4, 3, 7):
                lib.crct_lab_lnf_dbg(d)
                abl[d] = round(timed(lambda: ops.gemm(s_out, wf, M, N, H, bias=bvec, out=out, act="gelu", preact_out=pre, lnf=lnf)), 1)
            lib.crct_lab_lnf_dbg(0)
            print("   ablations (1 = no LN(s) write-back, 2 = no statistics loads, 4 = no epilogue correction):", abl)
        t_fold = timed(lambda: ops.ln_fold_weights(flat, [(0, N * H, N * H + N, N * H + N + H, H, N)], wfold=wfold), n=20)
        print("%-20s producer %5.1f us, with statistics %5.1f | LayerNorm %5.1f | consumer plain %5.1f, folded %5.1f | chain plain %5.1f, folded %5.1f | "
              "fold kernel (incl. host tables) %6.1f us for %.1f MB"
              % (name, t_prod, t_prod_s, t_ln, t_plain, t_lnf, t_prod + t_ln + t_plain, t_prod_s + t_lnf, t_fold, N * H * 6 / 1e6), flush=True)


if __name__ == "__main__":
    main()
