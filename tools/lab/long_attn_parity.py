"""Does the long-sequence attention path (attention_long.hip) cost gradient fidelity?  Full-depth step (vilbert.json, F_v = 1024,
dropout 0, seeded weights) against the fp32 CPU oracle and its bf16-autocast self (the yardstick of tests/test_step_gpu.py) on
padded batches: at T = 112 through the register-resident kernels AND through the long kernels (crct_attention_force_long), and
at T = 124 (long kernels only).  Prints min / p10 / median cosine and the norm-ratio range per case."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from crct import config as C, synthetic as S, lib as L        # noqa: E402
from crct.model import VisualDialogEncoder                     # noqa: E402
from crct.step_adapter import forward as step_forward          # noqa: E402
from oracle import crct_oracle as O                            # noqa: E402
from helpers import seeded_weights                             # noqa: E402


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def main():
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = C.vilbert_config(v_feature_size=1024, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
                           v_attention_probs_dropout_prob=0.0)
    params = C.default_params(device=torch.device("cuda"))
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.cls_dropout = 0.0
    S.seeded_fill_(model.state_dict(), base_seed=7)
    core._invalidate_shadow()
    model.train()
    cpu = dict(params, device=torch.device("cpu"))
    lib = L.load()
    args = sys.argv[1:]
    short = "short" in args          # "short": configs[1]'s lengths (20 tokens, 36 elements) instead, both kernel families
    seeds = [int(s) for s in args if s != "short"] or [1234, 1238]
    cases = ((112, [112, 64, 87, 99], [44, 29, 37, 44]), (124, [124, 71, 96, 110], [44, 29, 37, 44]), (124, [124] * 4, [44] * 4))
    if short:
        cases = ((20, [20, 17, 18, 16], [36, 30, 33, 36]),)
    for T, lens, nv in cases:
        for seed in seeds:
            batch = S.make_batch(4, T, max(nv), 1024, seed=seed, lengths=lens, n_vis=nv)
            batch["R"][:, 1] = torch.tensor([1.0, 1.0, 0.0, 1.0])
            batch["needs_reg"] = (batch["R"][:, 1:2] == 1)
            sd = seeded_weights(cfg, cpu, base_seed=7)
            ref = O.oracle_step(sd, cfg, cpu, batch, cls_dropout=0.0)
            ref[0].backward()
            sd16 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
            with torch.autocast("cpu", dtype=torch.bfloat16):
                r16 = O.oracle_step(sd16, cfg, cpu, batch, cls_dropout=0.0)
            r16[0].float().backward()
            keys = [k for k, v in sd.items() if v.grad is not None and float(v.grad.norm()) > 1e-7]

            def stats(get):
                cs = sorted(cosine(get(k), sd[k].grad) for k in keys)
                qs = [float(get(k).double().norm() / sd[k].grad.double().norm()) for k in keys]
                return "min %.4f p10 %.4f median %.4f norm %.3f..%.3f" % (cs[0], cs[len(cs) // 10], cs[len(cs) // 2], min(qs), max(qs))
            print("T=%d lens=%s seed=%d  loss oracle %.5f" % (T, lens, seed, float(ref[0])))
            print("   bf16-autocast oracle : " + stats(lambda k: sd16[k].grad.float()))
            modes = [(0, True), (1, True)] if (T <= 112 and not short) else [(0, True)]
            if short:
                modes = [(0, False), (0, True)]      # "short": the residual stream stored as bf16 (rounds 1 - 5) against fp32 (CrctStepCfg.residual_fp32)
            for force, r32 in modes:
                lib.crct_attention_force_long(force)
                core.residual_fp32 = r32
                core.zero_flat_grads()
                out = step_forward(model, batch, params)
                out[0].backward()
                torch.cuda.synchronize()
                named = dict(core.named_parameters())
                label = ("long kernels" if (force or T > 112) else "short kernels") + ("" if r32 else ", bf16 residual stream")
                print("   HIP %-38s: " % label + stats(lambda k: named[k].grad.float().cpu()) + "  loss %.5f" % float(out[0]))
            lib.crct_attention_force_long(0)
            core.residual_fp32 = True


if __name__ == "__main__":
    main()
