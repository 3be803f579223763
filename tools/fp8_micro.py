"""Developer micro-benchmark of the fp8 helper kernels (timing only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch
from crct import lib as L, ops
lib = L.load()
DEV = "cuda"
M, H, I = 1600, 768, 3072
x = torch.randn(M, H, device=DEV).to(torch.bfloat16)
gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
y = torch.empty_like(x); mean = torch.empty(M, device=DEV); rstd = torch.empty(M, device=DEV)
yq = torch.zeros(M, H, device=DEV, dtype=torch.uint8)
sc = torch.tensor([30.0], device=DEV); am = torch.zeros(L.FP8_AMAX_LANES, device=DEV)
def timeit(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-40s %8.2f us" % (name, (time.perf_counter() - t0) / n * 1e6), flush=True)
s = L.current_stream()
timeit("ln_fwd", lambda: lib.crct_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, H, 1e-12, 0, 1.0, 0, 0, s))
timeit("ln_fwd_q", lambda: lib.crct_layernorm_fwd_q(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, H, 1e-12, 0, 1.0, 0, 0, yq.data_ptr(), sc.data_ptr(), am.data_ptr(), s))
timeit("ln_fwd_q (no amax)", lambda: lib.crct_layernorm_fwd_q(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, H, 1e-12, 0, 1.0, 0, 0, yq.data_ptr(), sc.data_ptr(), None, s))
timeit("quantize_bf16", lambda: lib.crct_fp8_quantize_bf16(x.data_ptr(), yq.data_ptr(), sc.data_ptr(), am.data_ptr(), x.numel(), s))
timeit("quantize_bf16 (no amax)", lambda: lib.crct_fp8_quantize_bf16(x.data_ptr(), yq.data_ptr(), sc.data_ptr(), None, x.numel(), s))
w = (torch.randn(I, H, device=DEV) * 0.05)
xq = (x.float() * 30).clamp(-448, 448).to(torch.float8_e4m3fn)
wq = (w * 1000).clamp(-448, 448).to(torch.float8_e4m3fn)
sa, sb = torch.tensor([30.0], device=DEV), torch.tensor([1000.0], device=DEV)
b = torch.zeros(I, device=DEV)
pre = torch.empty(M, I, device=DEV, dtype=torch.bfloat16); hq = torch.zeros(M, I, device=DEV, dtype=torch.uint8)
timeit("gemm_fp8 plain", lambda: ops.gemm_fp8(xq, wq, sa, sb, M, I, H, bias=b))
timeit("gemm_fp8 gelu+preact", lambda: ops.gemm_fp8(xq, wq, sa, sb, M, I, H, bias=b, act="gelu", preact_out=pre))
timeit("gemm_fp8 gelu+preact+q_out", lambda: ops.gemm_fp8(xq, wq, sa, sb, M, I, H, bias=b, act="gelu", preact_out=pre, q_out=hq, q_scale=sc, q_amax=am))
timeit("gemm_fp8 gelu+preact+q_out(no amax)", lambda: ops.gemm_fp8(xq, wq, sa, sb, M, I, H, bias=b, act="gelu", preact_out=pre, q_out=hq, q_scale=sc, q_amax=None))
xb = x; wb = w.to(torch.bfloat16)
timeit("gemm bf16 gelu+preact", lambda: ops.gemm(xb, wb, M, I, H, bias=b, act="gelu", preact_out=pre))
