"""Developer probe: full-size bf16 gradient parity against the fp32 oracle over several batches / weight seeds (is a parity
number a property of the build or one draw of the bf16 rounding noise?).
    python tools/lab/parity_draws.py <tag>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from crct import config as C, synthetic as S
from crct.model import VisualDialogEncoder
from crct.step_adapter import forward as step_forward
from oracle import crct_oracle as O
from helpers import seeded_weights

dev = torch.device("cuda", 0)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
cpu_params = dict(C.default_params(), device=torch.device("cpu"))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


for wseed, bseed in ((11, 77), (11, 78), (12, 79), (13, 80)):
    batch = S.make_batch(80, 20, 36, 2048, seed=bseed)
    sd = seeded_weights(cfg, cpu_params, base_seed=wseed)
    ref = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
    ref[0].backward()
    params = C.default_params(device=dev)
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.cls_dropout = 0.0
    S.seeded_fill_(model.state_dict(), base_seed=wseed)
    core._invalidate_shadow()
    out = step_forward(model, batch, params)
    out[0].backward()
    torch.cuda.synchronize()
    named = dict(core.named_parameters())
    cs = sorted(cosine(named[k].grad.float().cpu(), sd[k].grad) for k in sd if sd[k].grad is not None and float(sd[k].grad.double().norm()) >= 1e-7)
    print("%s weights %d batch %d: loss %.5f (oracle %.5f)  gradient cosine min %.4f p10 %.4f median %.4f" %
          (sys.argv[1] if len(sys.argv) > 1 else "", wseed, bseed, float(out[0]), float(ref[0]), cs[0], cs[len(cs) // 10], cs[len(cs) // 2]))
    sys.stdout.flush()
    del model, core, named
    torch.cuda.empty_cache()
