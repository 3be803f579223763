"""Developer probe: the GELU / GELU' epilogues of the GEMM kernels against torch on GEMM outputs of growing magnitude (the visual
stream's pre-activations are larger than the text stream's)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch
import torch.nn.functional as F

from crct import ops

dev = "cuda"
torch.manual_seed(0)
for (M, N, K) in ((2880, 1024, 1024), (1600, 3072, 768)):
    for scale in (0.02, 0.1, 0.5, 2.0, 8.0):
        x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * scale / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev) * scale
        u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        h = ops.gemm(x, w, M, N, K, bias=b, act="gelu", preact_out=u)
        uf = x.float() @ w.float().t() + b
        ref = F.gelu(uf)
        err_h = (h.float() - ref).abs().max().item() / ref.abs().max().item()
        err_u = (u.float() - uf).abs().max().item() / uf.abs().max().item()
        # backward epilogue: dy W * gelu'(u)
        dy = torch.randn(M, K, device=dev).to(torch.bfloat16)
        g = ops.gemm(dy, w, M, N, K, dact_src=u, dact="gelu")
        uu = u.float().requires_grad_(True)
        F.gelu(uu).backward(dy.float() @ w.float().t())
        err_g = (g.float() - uu.grad).abs().max().item() / uu.grad.abs().max().item()
        bad = torch.isnan(h.float()).sum().item() + torch.isnan(g.float()).sum().item()
        print("M %d N %d K %d  |u| max %8.2f : gelu rel err %.2e  preact rel err %.2e  gelu' epilogue rel err %.2e  NaNs %d" %
              (M, N, K, uf.abs().max().item(), err_h, err_u, err_g, bad))
