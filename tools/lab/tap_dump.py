"""Developer probe: every per-layer hidden state (engine taps) of one full-size forward pass (B = 80, dropout 0, seeded weights) as a
checksum table + the raw taps in a .pt file, to compare two builds of the library layer by layer.
    python tools/lab/tap_dump.py <out.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT):
    sys.path.insert(0, p)
import torch

from crct import config as C, synthetic as S
from crct.model import VisualDialogEncoder
from crct.step_adapter import forward as step_forward

dev = torch.device("cuda", 0)
cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
params = C.default_params(device=dev)
model = VisualDialogEncoder(params, config=cfg)
core = model.bert_pretrained
core.cls_dropout = 0.0
S.seeded_fill_(model.state_dict(), base_seed=11)
core._invalidate_shadow()
B, T, V = 80, 20, 36
batch = S.make_batch(B, T, V, 2048, seed=77)
out = step_forward(model, batch, params, output_nsp_scores=True)
out[0].backward()
torch.cuda.synchronize()
eng = core._engine
names = ["emb.t", "emb.v"] + ["%s%d.%s" % (k, i, s) for (k, i) in __import__("crct.layout", fromlist=["x"]).encoder_schedule(cfg) for s in ("t", "v")] + ["seq_t", "seq_v"]
taps = {}
for n in names:
    try:
        taps[n] = eng.tap(n, B, T, V).float().cpu()
    except Exception as e:
        print("tap", n, "failed:", e)
small = {"loss": out[0].detach().float().cpu()}
for n, t in taps.items():
    small["tap." + n] = t.flatten()[::251].clone()
for e in core.table:
    if e.used:
        g = core.flat_grads[e.offset:e.offset + e.numel].detach().float().cpu()
        small["grad." + e.name] = (float(g.double().norm()), g[::max(1, e.numel // 4096)].clone())
torch.save(small, sys.argv[1])
for n, t in taps.items():
    print("%-8s sum|x| %.6e  max %.4f" % (n, t.abs().double().sum().item(), t.abs().max().item()))
