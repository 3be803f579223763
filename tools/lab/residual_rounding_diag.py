"""CPU diagnosis (no GPU): where does the HIP path's bf16 gradient noise beyond the bf16-autocast yardstick come from?
The oracle under torch's CPU bf16 autocast keeps the residual stream, LayerNorm and softmax in fp32 (only GEMM operands are rounded);
the HIP path also STORES the pre-LayerNorm sums and the LayerNorm outputs as bf16 (and their gradients on the way back).  This script
adds exactly those roundings to the autocast oracle (value rounded forward, gradient rounded backward, at the LayerNorm input and output)
and prints min / p10 / median gradient cosine against the fp32 oracle for: autocast, + both, + values only, + gradients only.  Same batches as
tools/lab/long_attn_parity.py short <seeds> (4 sequences, 20 tokens, 36 elements, full depth), whose HIP lines it is to be read beside.
    python tools/lab/residual_rounding_diag.py 1234 81 84"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from crct import config as C, synthetic as S        # noqa: E402
from oracle import crct_oracle as O                   # noqa: E402
from helpers import seeded_weights                    # noqa: E402


class RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd=True, bwd=True):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    cfg = C.vilbert_config(v_feature_size=1024, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
                           v_attention_probs_dropout_prob=0.0)
    cpu = dict(C.default_params(), device=torch.device("cpu"))
    seeds = [int(s) for s in sys.argv[1:]] or [1234, 81, 84]
    plain_ln = O.layer_norm

    def make_ln(fwd, bwd):
        def ln(x, w, b):
            return RoundBf16.apply(plain_ln(RoundBf16.apply(x.float(), fwd, bwd), w, b), fwd, bwd)
        return ln
    variants = (("autocast", plain_ln), ("+ residual stream bf16, values and gradients", make_ln(True, True)),
                ("+ values only (forward)", make_ln(True, False)), ("+ gradients only (backward)", make_ln(False, True)))
    totals = {k: [0.0, 0.0, 0.0] for k, _ in variants}
    for seed in seeds:
        batch = S.make_batch(4, 20, 36, 1024, seed=seed, lengths=[20, 17, 18, 16], n_vis=[36, 30, 33, 36])
        batch["R"][:, 1] = torch.tensor([1.0, 1.0, 0.0, 1.0])
        batch["needs_reg"] = (batch["R"][:, 1:2] == 1)
        sd = seeded_weights(cfg, cpu, base_seed=7)
        ref = O.oracle_step(sd, cfg, cpu, batch, cls_dropout=0.0)
        ref[0].backward()
        keys = [k for k, v in sd.items() if v.grad is not None and float(v.grad.norm()) > 1e-7]
        print("seed %d  loss %.5f" % (seed, float(ref[0])), flush=True)
        for label, ln in variants:
            s16 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
            O.layer_norm = ln
            try:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    r = O.oracle_step(s16, cfg, cpu, batch, cls_dropout=0.0)
                r[0].float().backward()
            finally:
                O.layer_norm = plain_ln
            cs = sorted(cosine(s16[k].grad.float(), sd[k].grad) for k in keys)
            print("   %-46s min %.4f p10 %.4f median %.4f   loss %.5f" % (label, cs[0], cs[len(cs) // 10], cs[len(cs) // 2], float(r[0])), flush=True)
            for i, v in enumerate((cs[0], cs[len(cs) // 10], cs[len(cs) // 2])):
                totals[label][i] += (1.0 - v) / len(seeds)
    print("mean deficit (1 - cosine) over %d draws:" % len(seeds))
    for label, _ in variants:
        print("   %-46s min %.4f p10 %.4f median %.4f" % ((label,) + tuple(totals[label])))


if __name__ == "__main__":
    main()
