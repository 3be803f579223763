"""Host-side cost of one training step: cProfile over N steps of the configs[1] loop (the GPU runs behind; the numbers are host time).
    python tools/lab/host_profile.py [tree_root]      # tree_root: another checkout of the repository (default: this one)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
T = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0, "rest": 0.0}


def step(i):
    t0 = time.perf_counter()
    loss = step_forward(model, pool[i % 8], params)[0]
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    opt.zero_grad()
    sched.step()
    t4 = time.perf_counter()
    T["fwd"] += t1 - t0; T["bwd"] += t2 - t1; T["opt"] += t3 - t2; T["rest"] += t4 - t3


for i in range(10):
    step(i)
torch.cuda.synchronize()
for k in T:
    T[k] = 0.0
N = 40
t0 = time.perf_counter()
for i in range(N):
    step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print("tree %s: host %.3f ms per step (forward %.3f, backward %.3f, optimizer %.3f, zero_grad + scheduler %.3f); with the GPU drained %.3f ms per step"
      % (ROOT, host / N * 1e3, T["fwd"] / N * 1e3, T["bwd"] / N * 1e3, T["opt"] / N * 1e3, T["rest"] / N * 1e3, total / N * 1e3), flush=True)
# synchronised per phase: the host's own cost, nothing blocking on a full queue
for k in T:
    T[k] = 0.0
for i in range(N):
    torch.cuda.synchronize()
    step(i)
torch.cuda.synchronize()
print("synchronised steps: forward %.3f, backward %.3f, optimizer %.3f, rest %.3f ms of host time" % tuple(T[k] / N * 1e3 for k in ("fwd", "bwd", "opt", "rest")), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    torch.cuda.synchronize()
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(14)
