"""Developer probe (1 GPU): the data-parallel call pattern on a single-rank RCCL communicator.
Checks that the bucketed exchange (event mode and segment mode) leaves results bit-identical to the plain step
and times the three variants.  Not part of the product or tests."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch
import torch.distributed as dist

from crct import config as CFG, synthetic as S
from crct.model import VisualDialogEncoder
from crct.optim import get_optimizer
from crct.step_adapter import forward as step_forward
from crct.ddp import FlatGradDDP

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(mode, steps=40, B=80, warm=10):
    cfg = CFG.vilbert_config(v_feature_size=2048)
    params = CFG.default_params(device=dev, rank=0, world_size=1, ddp=True, batch_size=B, seed=0)
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.sync_stats = False
    model.train()
    opt = get_optimizer(params, model)
    opt.overlap = True
    if mode != "plain":
        ddp = FlatGradDDP(model, bucket_mb=64)
        ddp.force_exchange = True
        ddp.event_mode = mode == "events"
    pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.make_batch(B, 20, 36, 2048, seed=1234 + 97 * i).items()} for i in range(4)]
    t0 = None
    g1 = None
    for it in range(steps):
        if it == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = step_forward(model, pool[it % 4], params)[0]
        loss.backward()
        if it == 0:
            torch.cuda.synchronize()
            g1 = core.flat_grads.clone()
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps - warm) * 1e3
    opt.synchronize()
    return g1, float(loss), dt


_, _, t_before = run("plain")
print("plain step before the RCCL communicator exists: %.2f ms" % t_before)
if os.environ.get("PROBE_LAZY"):
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
else:
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
ref, l0, t_plain = run("plain")
for mode in ("events", "segments"):
    p, l, t = run(mode)
    rel = float((p - ref).abs().max() / ref.abs().max())
    print("%-9s loss %.6f (plain %.6f)  step-0 gradients: max |dg| / max |g| = %.3e  %.2f ms/step (plain %.2f)" % (mode, l, l0, rel, t, t_plain))
dist.destroy_process_group()
