"""Developer probe: what does the vendor GEMM (torch.mm -> hipBLASLt) reach at the CRCT shapes?  Run under
rocprofv3 --kernel-trace --stats; kernel names carry the macro-tile.  Not part of the product or tests."""
import torch
dev = "cuda"
shapes = [("t.qkv fwd", 1600, 2304, 768), ("t.ffn_up fwd", 1600, 3072, 768), ("t.ffn_dn fwd", 1600, 768, 3072), ("t.out fwd", 1600, 768, 768),
          ("v.qkv fwd", 2880, 3072, 1024), ("v.ffn fwd", 2880, 1024, 1024), ("v.emb fwd", 2880, 1024, 2048)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    wt = w.t().contiguous()
    dy = torch.randn(M, N, device=dev).bfloat16()
    for tag, fn in (("fwd  x@w.T", lambda: torch.mm(x, w.t())), ("dgrad dy@w", lambda: torch.mm(dy, w)), ("wgrad dy.T@x", lambda: torch.mm(dy.t(), x))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print("%-14s %-12s M=%5d N=%5d K=%5d  %7.2f us  %7.1f TF" % (name, tag, M, N, K, us, 2.0 * M * N * K / us / 1e6))
