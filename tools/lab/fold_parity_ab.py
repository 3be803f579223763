"""Gradient parity against the fp32 oracle with the LayerNorm forward folded into the GEMMs and as launches of its own, same weights and
batch (tests/test_step_gpu.py::test_edge_shapes_match_oracle / test_full_size_step_matches_oracle shapes):
    python tools/lab/fold_parity_ab.py B V T [seed ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch                                   # noqa: E402
from crct import config as C, synthetic as S   # noqa: E402
from crct.model import VisualDialogEncoder     # noqa: E402
from crct.step_adapter import forward as step_forward   # noqa: E402
from oracle import crct_oracle as O            # noqa: E402
from helpers import seeded_weights             # noqa: E402


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def main():
    B, V, T = (int(x) for x in sys.argv[1:4])
    seeds = [int(x) for x in sys.argv[4:]] or [11]
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                           v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
    for seed in seeds:
        batch = S.make_batch(B, T, V, 2048, seed=100 + B + seed)
        cpu_params = dict(C.default_params(), device=torch.device("cpu"))
        sd = seeded_weights(cfg, cpu_params, base_seed=seed)
        ref = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
        ref[0].backward()
        for fold in (True, False):
            params = dict(C.default_params(), device=torch.device("cuda"), ln_fold=fold)
            model = VisualDialogEncoder(params, config=cfg)
            core = model.bert_pretrained
            core.cls_dropout = 0.0
            S.seeded_fill_(model.state_dict(), base_seed=seed)
            core._invalidate_shadow()
            out = step_forward(model, batch, params, output_nsp_scores=True)
            out[0].backward()
            torch.cuda.synchronize()
            rows = []
            for k, p in core.named_parameters():
                r = sd[k].grad
                if r is None or float(r.double().norm()) < 1e-7:
                    continue
                g = p.grad.float().cpu()
                rows.append((cosine(g, r), float(g.double().norm()) / float(r.double().norm()), k))
            cos = sorted(c for c, _, _ in rows)
            print("seed %d fold %-5s loss %.6f (oracle %.6f)  cosine min %.4f p10 %.4f median %.4f  ratio %.3f .. %.3f  worst %s"
                  % (seed, fold, float(out[0]), float(ref[0]), cos[0], cos[len(cos) // 10], cos[len(cos) // 2], min(q for _, q, _ in rows),
                     max(q for _, q, _ in rows), [(k.replace("bert.encoder.", ""), round(c, 3)) for c, _, k in sorted(rows)[:2]]), flush=True)
            del model, core


if __name__ == "__main__":
    main()
