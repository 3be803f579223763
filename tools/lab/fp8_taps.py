"""Per-layer hidden states of the fp8 forward against the fp32 oracle and its e4m3 emulation (developer tooling)."""
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


def build_model(cfg, params, weights=None, seed=7):
    params = dict(params, device=torch.device("cuda:0"))
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.cls_dropout = 0.0
    S.seeded_fill_(model.state_dict(), base_seed=seed)
    core._invalidate_shadow()
    return model, params

cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
B, T, V = 8, 20, 36
batch = S.make_batch(B, T, V, 2048, seed=31)
cpu_params = dict(C.default_params(), device=torch.device("cpu"))
torch.set_num_threads(16)
taps32, taps8 = {}, {}
sd = seeded_weights(cfg, cpu_params, base_seed=11)
O.oracle_step(sd, cfg, cpu_params, batch, taps=taps32, cls_dropout=0.0)
O.FP8_EMULATION = True
sd8 = seeded_weights(cfg, cpu_params, base_seed=11)
O.oracle_step(sd8, cfg, cpu_params, batch, taps=taps8, cls_dropout=0.0)
O.FP8_EMULATION = False
model, params = build_model(cfg, C.default_params(fp8=True), weights=None, seed=11)
core = model.bert_pretrained
out = step_forward(model, batch, params, output_nsp_scores=True)
torch.cuda.synchronize()
for name, ref in taps32.items():
    if name == "reg_raw":
        continue
    try:
        got = core._engine.tap(name, B, T, V).float().cpu()
    except RuntimeError:
        continue
    r = ref.detach()
    e_hip = float((got - r).norm() / (r.norm() + 1e-9))
    e_emu = float((taps8[name].detach() - r).norm() / (r.norm() + 1e-9))
    print("%-16s rel l2 error vs fp32: product %.4f   emulation %.4f" % (name, e_hip, e_emu))

for name in ("t0.t", "t5.t", "t11.t", "seq_t"):
    if name not in taps32:
        continue
    r = taps32[name].detach(); g = core._engine.tap(name, B, T, V).float().cpu(); m = taps8[name].detach()
    r0, g0, m0 = r[:, 0], g[:, 0], m[:, 0]
    print("%-8s CLS rows: product %.4f emulation %.4f | other rows: product %.4f emulation %.4f | product-vs-emulation all %.4f" % (
        name, float((g0 - r0).norm() / r0.norm()), float((m0 - r0).norm() / r0.norm()),
        float((g[:, 1:] - r[:, 1:]).norm() / r[:, 1:].norm()), float((m[:, 1:] - r[:, 1:]).norm() / r[:, 1:].norm()), float((g - m).norm() / m.norm())))

out[0].backward()
torch.cuda.synchronize()
ref = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
ref[0].backward()
rows = []
for k, p in core.named_parameters():
    r = sd[k].grad
    if r is None or float(r.double().norm()) < 1e-7:
        continue
    a, b = p.grad.float().cpu().double().flatten(), r.double().flatten()
    rows.append((float((a @ b) / (a.norm() * b.norm() + 1e-300)), k))
cos = sorted(c for c, _ in rows)
print("first-pass gradient parity", os.environ.get("LAB_NO_CTXQ"), os.environ.get("LAB_NO_DQKVQ"), dict(min=round(cos[0], 4), p10=round(cos[len(cos) // 10], 4), median=round(cos[len(cos) // 2], 4)), sorted(rows)[:3])
print("loss", float(out[0]), float(ref[0]))

O.FP8_EMULATION = True
O.oracle_step(sd8, cfg, cpu_params, batch, cls_dropout=0.0)[0].backward()
O.FP8_EMULATION = False
def cosv(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))
named = dict(core.named_parameters())
for c, k in sorted(rows)[:12]:
    print("%-60s product %.4f  emulation %.4f  product-vs-emulation %.4f" % (k, c, cosv(sd8[k].grad, sd[k].grad), cosv(named[k].grad.float().cpu(), sd8[k].grad)))
