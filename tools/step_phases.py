"""Developer probe: GPU-side duration of the phases of a training step WITHOUT a profiler attached
(torch events on the caller's stream), next to the host time spent enqueuing each phase."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch

from crct import config as CFG, synthetic as S
from crct.model import VisualDialogEncoder
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero
from crct.step_adapter import forward as step_forward

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--exchange", choices=("none", "bf16", "fp32"), default="none", help="run the data-parallel exchange on a single-rank RCCL communicator")
ap.add_argument("--thread", type=int, default=1)
ap.add_argument("--dtype", choices=("bf16", "fp8"), default="bf16", help="fp8: BASELINE configs[4] (every encoder GEMM on fp8 operands)")
ap.add_argument("--fp8-bf16-forward", action="store_true", help="--dtype fp8 with bf16 forward GEMMs (params['fp8_forward'] = False)")
ap.add_argument("--stats", type=int, default=0, help="1: the asynchronous 9-float stats all-reduce of bench.py inside the step")
args = ap.parse_args()
dev = torch.device("cuda", 0)
B = 80
cfg = CFG.vilbert_config(v_feature_size=2048)
params = CFG.default_params(device=dev, rank=0, world_size=1, ddp=False, batch_size=B, seed=0, fp8=args.dtype == "fp8")
if args.fp8_bf16_forward:
    params["fp8_forward"] = False
model = VisualDialogEncoder(params, config=cfg)
core = model.bert_pretrained
core.sync_stats = False
model.train()
opt = get_optimizer(params, model)
opt.overlap = True
sched = WarmupLinearScheduleNonZero(opt, warmup_steps=params["warmup"], t_total=60000, min_lr=params["min_lr"])
pool = [{k: v.to(dev) for k, v in S.make_batch(B, 20, 36, 2048, seed=1234 + 97 * i).items()} for i in range(8)]
launch_host = []
if args.exchange != "none":
    import torch.distributed as dist
    from crct import ddp as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    ddp = D.FlatGradDDP(model, grad_dtype=torch.bfloat16 if args.exchange == "bf16" else torch.float32)
    ddp.force_exchange = True
    if hasattr(ddp, "use_thread"):
        ddp.use_thread = bool(args.thread)
    stats_red = D.AsyncStats(1, device=dev) if args.stats else None
    _orig_launch = D.BucketExchange.launch

    def timed_launch(self, b):
        t0 = time.perf_counter()
        _orig_launch(self, b)
        launch_host.append((time.perf_counter() - t0) * 1e3)
    D.BucketExchange.launch = timed_launch

marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, ev, time.perf_counter()))


orig_bwd = core._run_backward


def run_backward(tensors, step):
    mark("bwd_enter")
    orig_bwd(tensors, step)
    mark("bwd_enqueued")


core._run_backward = run_backward
orig_fwd = core._engine.forward if core._engine else None

N, W = 30, 10
rows = []
for it in range(N):
    marks.clear()
    mark("step_begin")
    out = step_forward(model, pool[it % 8], params)
    mark("fwd_loss_enqueued")
    if args.exchange != "none" and stats_red is not None:
        stats_red.launch(core.last_stats[8:17])
    out[0].backward()
    mark("autograd_done")
    opt.step()
    opt.zero_grad()
    sched.step()
    if args.exchange != "none" and stats_red is not None:
        stats_red.result()
    mark("opt_enqueued")
    if it >= W:
        torch.cuda.synchronize()
        base_ev, base_t = marks[0][1], marks[0][2]
        rows.append([(n, base_ev.elapsed_time(ev), (t - base_t) * 1e3) for n, ev, t in marks])
torch.cuda.synchronize()
print("synchronised every step (host starts each step with an empty queue): GPU ms since step_begin | host ms")
for i, (n, _, _) in enumerate(rows[0]):
    g = sum(r[i][1] for r in rows) / len(rows)
    h = sum(r[i][2] for r in rows) / len(rows)
    print("  %-20s gpu %7.3f   host %7.3f" % (n, g, h))

# free-running: events only, no per-step sync
allm = []
for it in range(N):
    marks.clear()
    mark("step_begin")
    out = step_forward(model, pool[it % 8], params)
    mark("fwd_loss_enqueued")
    if args.exchange != "none" and stats_red is not None:
        stats_red.launch(core.last_stats[8:17])
    out[0].backward()
    mark("autograd_done")
    opt.step()
    opt.zero_grad()
    sched.step()
    if args.exchange != "none" and stats_red is not None:
        stats_red.result()
    mark("opt_enqueued")
    allm.append(list(marks))
torch.cuda.synchronize()
print("free running: GPU ms since that step's step_begin | host ms since that step's begin; step period = gpu(next step_begin)")
keep = allm[W:-1]
for i, (n, _, _) in enumerate(keep[0]):
    g = sum(m[0][1].elapsed_time(m[i][1]) for m in keep) / len(keep)
    h = sum((m[i][2] - m[0][2]) * 1e3 for m in keep) / len(keep)
    print("  %-20s gpu %7.3f   host %7.3f" % (n, g, h))
per = sum(allm[k][0][1].elapsed_time(allm[k + 1][0][1]) for k in range(W, N - 1)) / (N - 1 - W)
hper = sum((allm[k + 1][0][2] - allm[k][0][2]) * 1e3 for k in range(W, N - 1)) / (N - 1 - W)
print("  step period: gpu %.3f ms, host %.3f ms" % (per, hper))
if launch_host:
    print("  host time inside BucketExchange.launch: %.3f ms per launch, %.3f ms per step (%d launches per step)"
          % (sum(launch_host) / len(launch_host), sum(launch_host) / (2 * N), len(launch_host) // (2 * N)))
    dist.destroy_process_group()
