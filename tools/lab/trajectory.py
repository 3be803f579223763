"""Loss trajectory of a short training run on a fixed pool of 8 synthetic batches (the bench workload, dropout 0.1): bf16 against the
fp8 mode (configs[4]).  The pool is small enough to be learnt, so the loss falls; printed as the mean over every 50 steps.
Developer tooling (one process per dtype: python tools/lab/trajectory.py bf16|fp8 [steps] [plotqa-real]; the third argument runs the
reference's own PlotQA shape -- B 80, 44 elements x 1024-d, 124 tokens with the padding real samples carry (lengths 60 .. 124) -- through the
long-sequence attention kernels)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch

from crct import config as CFG, synthetic as S
from crct.model import VisualDialogEncoder
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero
from crct.step_adapter import forward as step_forward

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
B = 80
real = len(sys.argv) > 3 and sys.argv[3] == "plotqa-real"
T, V, F = (124, 44, 1024) if real else (20, 36, 2048)
cfg = CFG.vilbert_config(v_feature_size=F)
params = CFG.default_params(device=dev, rank=0, world_size=1, ddp=False, batch_size=B, seed=0, fp8=dtype == "fp8")
model = VisualDialogEncoder(params, config=cfg)
core = model.bert_pretrained
core.sync_stats = False
model.train()
opt = get_optimizer(params, model)
opt.overlap = True
sched = WarmupLinearScheduleNonZero(opt, warmup_steps=20, t_total=steps, min_lr=params["min_lr"])
g = torch.Generator().manual_seed(5)
extra = [dict(lengths=torch.randint(60, T + 1, (B,), generator=g).tolist(), n_vis=torch.randint(20, V + 1, (B,), generator=g).tolist()) if real else {} for _ in range(8)]
pool = [{k: v.to(dev) for k, v in S.make_batch(B, T, V, F, seed=1234 + 97 * i, **extra[i]).items()} for i in range(8)]
losses = []
for it in range(steps):
    loss = step_forward(model, pool[it % 8], params)[0]
    loss.backward()
    opt.step()
    opt.zero_grad()
    sched.step()
    losses.append(loss.detach())
torch.cuda.synchronize()
vals = torch.stack(losses).float().cpu()
assert bool(torch.isfinite(vals).all())
print(dtype, "mean loss per 50 steps:", " ".join("%.4f" % float(vals[i:i + 50].mean()) for i in range(0, steps, 50)))
