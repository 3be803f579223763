"""Developer timing: forward / backward / optimizer wall time per step with device syncs in between."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
import torch
from crct import config as CFG, synthetic as S
from crct.model import VisualDialogEncoder
from crct.optim import get_optimizer
from crct.step_adapter import forward as step_forward
dev = torch.device("cuda:0")
cfg = CFG.vilbert_config(v_feature_size=2048)
params = CFG.default_params(device=dev)
model = VisualDialogEncoder(params, config=cfg); core = model.bert_pretrained; core.sync_stats = False
opt = get_optimizer(params, model)
pool = [{k: v.to(dev) for k, v in S.make_batch(80, 20, 36, 2048, seed=i).items()} for i in range(4)]
def sync(): torch.cuda.synchronize()
for it in range(5):
    out = step_forward(model, pool[it % 4], params); out[0].backward(); opt.step(); opt.zero_grad()
sync()
tf = tb = to = thost_f = thost_b = 0.0
N = 20
for it in range(N):
    t0 = time.perf_counter(); out = step_forward(model, pool[it % 4], params); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    out[0].backward(); t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
    opt.step(); opt.zero_grad(); sync(); t5 = time.perf_counter()
    tf += t2 - t0; tb += t4 - t2; to += t5 - t4; thost_f += t1 - t0; thost_b += t3 - t2
print("forward %.2f ms (host enqueue %.2f)  backward %.2f ms (host enqueue %.2f)  optimizer %.2f ms" % (tf / N * 1e3, thost_f / N * 1e3, tb / N * 1e3, thost_b / N * 1e3, to / N * 1e3))
