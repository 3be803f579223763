"""Developer probe (TIMING ONLY, wrong numbers): the training step of bench.py with the GEMM main loops ablated by the LAB build of the
library (tools/lab/libcrct_hip.so, -DCRCT_GEMM_LAB; CRCT_GEMM_DBG bits: 1 = no epilogue, 2 = no operand DMA after the prologue).
What does the step cost when the GEMMs move no operand bytes / write no results?  Run through tools/lab/step_ablate.sh, which puts the
LAB library in the package's place for the duration.

    CRCT_GEMM_DBG=2 python tools/lab/step_ablate.py [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
import torch

from crct import config as CFG, synthetic as S
from crct.model import VisualDialogEncoder
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero
from crct.step_adapter import forward as step_forward

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
cfg = CFG.vilbert_config(v_feature_size=2048)
params = CFG.default_params(device=dev, batch_size=80, seed=0)
model = VisualDialogEncoder(params, config=cfg)
core = model.bert_pretrained
core.sync_stats = False
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


for i in range(8):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print("CRCT_GEMM_DBG=%s  %.3f ms per step" % (os.environ.get("CRCT_GEMM_DBG", "0"), (time.perf_counter() - t0) / steps * 1e3))
