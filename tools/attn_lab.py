"""Developer timing of the attention kernels at the CRCT shapes (not part of the product or tests)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch
from crct import ops
dev = "cuda"
shapes = [("text self", 80, 16, 20, 20, 48), ("visual self", 80, 16, 36, 36, 64), ("co t->v", 80, 32, 20, 36, 32), ("co v->t", 80, 32, 36, 20, 32),
          ("visual self long", 64, 16, 100, 100, 64), ("text self long", 64, 16, 40, 40, 48)]
for name, B, h, Tq, Tk, d in shapes:
    Hh = h * d
    q = torch.randn(B, Tq, 3 * Hh, device=dev).bfloat16(); k = torch.randn(B, Tk, 3 * Hh, device=dev).bfloat16()
    km = torch.ones(B, Tk, dtype=torch.uint8, device=dev)
    do = torch.randn(B, Tq, Hh, device=dev).bfloat16()
    for p in (0.0, 0.1):
        f = lambda: ops.attention_fwd(q[:, :, :Hh], k[:, :, Hh:2 * Hh], k[:, :, 2 * Hh:], km, h, d, p_drop=p, site=1, seed=5)
        b = lambda: ops.attention_bwd(q[:, :, :Hh], k[:, :, Hh:2 * Hh], k[:, :, 2 * Hh:], km, do, h, d, p_drop=p, site=1, seed=5)
        res = []
        for fn in (f, b):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print("%-18s B=%d h=%d Tq=%d Tk=%d d=%d p=%.1f  fwd %7.1f us  bwd %7.1f us" % (name, B, h, Tq, Tk, d, p, res[0], res[1]))
