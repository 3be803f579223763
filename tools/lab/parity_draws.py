"""Developer probe: full-size bf16 gradient parity against the fp32 oracle over several (weights, batch) draws, next to what makes a draw
hard -- how far the per-row gradients cancel in the batch sum (round 5: the mechanism behind "a parity number is one draw"):
    python tools/lab/parity_draws.py <tag> [B V T] [wseed:bseed ...]
Per draw: this path's gradient cosine against the fp32 oracle (min / p10 / median over the tensors), the same for the oracle under torch's
CPU bf16 autocast (the yardstick), the norm of the total gradient, and the coherence |sum_b g_b| / sum_b |g_b| of the per-row upstream
gradients at the CLS / IMG rows."""
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


args = sys.argv[2:]
B, V, T = (int(x) for x in args[:3]) if len(args) >= 3 and ":" not in args[0] else (80, 36, 20)
draws = [tuple(int(v) for v in x.split(":")) for x in args if ":" in x] or [(11, 77), (11, 78), (11, 81), (11, 82)]
for wseed, bseed in draws:
    batch = S.make_batch(B, T, V, 2048, seed=bseed)
    sd = seeded_weights(cfg, cpu_params, base_seed=wseed)
    taps = {}
    ref = O.oracle_step(sd, cfg, cpu_params, batch, taps=taps, cls_dropout=0.0)
    for k in ("seq_t", "seq_v"):
        taps[k].retain_grad()
    ref[0].backward()
    coh = [float(taps[k].grad[:, 0].sum(0).norm() / taps[k].grad[:, 0].norm(dim=1).sum()) for k in ("seq_t", "seq_v")]
    gnorm = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
    keys = [k for k in sd if sd[k].grad is not None and float(sd[k].grad.double().norm()) >= 1e-7]
    sd16 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        r16 = O.oracle_step(sd16, cfg, cpu_params, batch, cls_dropout=0.0)
    r16[0].float().backward()
    yard = sorted(cosine(sd16[k].grad.float(), sd[k].grad) for k in keys)
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
    cs = sorted(cosine(named[k].grad.float().cpu(), sd[k].grad) for k in keys)
    print("%s (%d,%d,%d) weights %d batch %d: loss %.5f (oracle %.5f)  gradient cosine min %.4f p10 %.4f median %.4f | bf16-autocast oracle min %.4f p10 %.4f "
          "median %.4f | total gradient norm %.3f, per-row coherence text %.3f visual %.3f" %
          (sys.argv[1] if len(sys.argv) > 1 else "", B, V, T, wseed, bseed, float(out[0]), float(ref[0]), cs[0], cs[len(cs) // 10], cs[len(cs) // 2],
           yard[0], yard[len(yard) // 10], yard[len(yard) // 2], gnorm, coh[0], coh[1]))
    sys.stdout.flush()
    del model, core, named
    torch.cuda.empty_cache()
