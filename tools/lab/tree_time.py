"""Step time (configs[1]) of another checkout of the repository: python tools/lab/tree_time.py <tree_root> -- for same-box A/Bs of commits."""
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1])
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
import torch                                             # noqa: E402
from crct import config as CFG, synthetic as S           # noqa: E402
from crct.model import VisualDialogEncoder               # noqa: E402
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero   # noqa: E402
from crct.step_adapter import forward as step_forward    # noqa: E402

dev = torch.device("cuda", 0)
cfg = CFG.vilbert_config(v_feature_size=2048)
params = CFG.default_params(device=dev, batch_size=80, seed=0)
model = VisualDialogEncoder(params, config=cfg)
core = model.bert_pretrained
core.sync_stats = False
core.stream_mode = (1, 1)
model.train()
opt = get_optimizer(params, model)
opt.overlap = True
sched = WarmupLinearScheduleNonZero(opt, warmup_steps=params["warmup"], t_total=60000, min_lr=params["min_lr"])
pool = [{k: v.to(dev) for k, v in S.make_batch(80, 20, 36, 2048, seed=1234 + 97 * i).items()} for i in range(8)]


def step(i):
    loss = step_forward(model, pool[i % 8], params)[0]
    loss.backward()
    opt.step()
    opt.zero_grad()
    sched.step()


for i in range(10):
    step(i)
out = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 30 * 1e3)
print("%s: %s ms per step" % (os.path.basename(ROOT) or "HEAD", ", ".join("%.3f" % x for x in out)), flush=True)
