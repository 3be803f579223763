"""Developer probe (CPU): what gradient fidelity does a block-scaled (MX: e4m3 + one e8m0 scale per 32 elements along
the contraction index -- the operand format of v_mfma_scale_f32_16x16x128_f8f6f4) fp8 step reach against the fp32 oracle, next to the
per-tensor e4m3 / e5m2 recipe the product runs today?  Oracle emulation only (oracle/crct_oracle.py, FP8_MX); full vilbert.json, B = 8.

    python tools/lab/mx_emulation.py [B] [seeds]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

from crct import config as C, synthetic as S
from oracle import crct_oracle as O
from helpers import seeded_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seeds = [int(s) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [31]
torch.set_num_threads(min(os.cpu_count() or 1, 16))
cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
cpu_params = dict(C.default_params(), device=torch.device("cpu"))


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def run(batch, fwd, bwd, wg, mx, fwd_bf16=False):
    sd = seeded_weights(cfg, cpu_params, base_seed=11)
    O.FP8_EMULATION, O.FP8_BWD_EMULATION, O.FP8_WGRAD_EMULATION, O.FP8_MX, O.FP8_FWD_BF16 = fwd, bwd, wg, mx, fwd_bf16
    try:
        out = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
        out[0].backward()
    finally:
        O.FP8_EMULATION = O.FP8_BWD_EMULATION = O.FP8_WGRAD_EMULATION = O.FP8_MX = O.FP8_FWD_BF16 = False
    return float(out[0]), sd


for seed in seeds:
    batch = S.make_batch(B, 20, 36, 2048, seed=seed)
    t0 = time.time()
    loss0, ref = run(batch, False, False, False, False)
    keys = [k for k in ref if ref[k].grad is not None and float(ref[k].grad.double().norm()) >= 1e-7]
    print("batch seed %d, B = %d: fp32 loss %.5f (%.0f s per oracle pass)" % (seed, B, loss0, time.time() - t0))
    for name, flags in (("per-tensor e4m3 fwd", (True, False, False, False)),
                        ("per-tensor fwd + e5m2 dgrad", (True, True, False, False)),
                        ("per-tensor fwd + dgrad + wgrad", (True, True, True, False)),
                        ("MX e4m3 fwd", (True, False, False, True)),
                        ("MX e4m3 fwd + dgrad", (True, True, False, True)),
                        ("MX e4m3 fwd + dgrad + wgrad", (True, True, True, True)),
                        ("fp32 fwd, per-tensor e5m2 dgrad", (True, True, False, False, True)),
                        ("fp32 fwd, per-tensor dgrad + wgrad", (True, True, True, False, True)),
                        ("fp32 fwd, MX e4m3 dgrad + wgrad", (True, True, True, True, True))):
        loss, sd = run(batch, *flags)
        cs = sorted(cosine(sd[k].grad, ref[k].grad) for k in keys)
        print("  %-34s loss %.5f (rel %.1e)   gradient cosine min %.4f  p10 %.4f  median %.4f" %
              (name, loss, abs(loss - loss0) / abs(loss0), cs[0], cs[len(cs) // 10], cs[len(cs) // 2]))
        sys.stdout.flush()
