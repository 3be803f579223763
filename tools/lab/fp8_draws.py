"""Is the product's fp8 gradient parity within the spread of the emulation's?  Several batches (draws), per draw the min / median
cosine of product and emulation gradients against the fp32 oracle (developer tooling)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd")); sys.path.insert(0, ROOT)
import torch
from helpers import seeded_weights
from crct import config as C
from crct import synthetic as S
from crct.model import VisualDialogEncoder
from crct.step_adapter import forward as step_forward
from oracle import crct_oracle as O

cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
B, T, V = int(os.environ.get("LAB_B", 8)), 20, 36
cpu_params = dict(C.default_params(), device=torch.device("cpu"))
torch.set_num_threads(16)


def cosv(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def summary(rows):
    c = sorted(rows)
    return "min %.4f p10 %.4f median %.4f" % (c[0], c[len(c) // 10], c[len(c) // 2])


for seed in (31, 32, 33, 34):
    batch = S.make_batch(B, T, V, 2048, seed=seed)
    sd = seeded_weights(cfg, cpu_params, base_seed=11)
    O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)[0].backward()
    keys = [k for k in sd if sd[k].grad is not None and float(sd[k].grad.double().norm()) >= 1e-7]
    res = {}
    for label, flags in (("emu fwd", (True, False, False)), ("emu fwd+bwd+wgrad", (True, True, True))):
        O.FP8_EMULATION, O.FP8_BWD_EMULATION, O.FP8_WGRAD_EMULATION = flags
        s8 = seeded_weights(cfg, cpu_params, base_seed=11)
        O.oracle_step(s8, cfg, cpu_params, batch, cls_dropout=0.0)[0].backward()
        O.FP8_EMULATION = O.FP8_BWD_EMULATION = O.FP8_WGRAD_EMULATION = False
        res[label] = summary([cosv(s8[k].grad, sd[k].grad) for k in keys])
    params = dict(C.default_params(fp8=True), device=torch.device("cuda:0"))
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.cls_dropout = 0.0
    S.seeded_fill_(model.state_dict(), base_seed=11)
    core._invalidate_shadow()
    named = dict(core.named_parameters())
    for label in ("hip fwd (calibration backward)", "hip fwd+bwd+wgrad"):
        core.zero_flat_grads()
        core._calls = 0
        out = step_forward(model, batch, params, output_nsp_scores=True)
        out[0].backward()
        torch.cuda.synchronize()
        res[label] = summary([cosv(named[k].grad.float().cpu(), sd[k].grad) for k in keys])
    print("batch seed %d:" % seed)
    for k, v in res.items():
        print("   %-32s %s" % (k, v))
