"""-m gpu: multi-step TRAJECTORY parity.  The single-step tests bound each gradient against the oracle; here the
whole loop of CRCT/train.py:205-215 -- forward, backward, AdamW (utils.py:228-249: one group per tensor, no weight decay on biases /
LayerNorm), warm-up + linear-decay schedule floored at min_lr (utils.py:11-29) -- runs for many steps on the HIP path (FusedAdamW,
one kernel) and on the fp32 CPU oracle (torch.optim.AdamW's arithmetic restated per tensor, oracle/crct_oracle.py) from the same
seeded weights over the same pool of batches, dropout 0.  Asserted: the loss curve pointwise within 2 %, every parameter tensor's
final value at cosine >= 0.999 of the oracle's, and -- the sensitive measure, since parameters barely move in a few dozen steps of
lr 2e-5 -- the cosine of the UPDATE each tensor received (final - initial), median and 10th percentile over the tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crct import config as C                                 # noqa: E402
from crct import synthetic as S                              # noqa: E402
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero   # noqa: E402
from crct.step_adapter import forward as step_forward        # noqa: E402
from oracle import crct_oracle as O                          # noqa: E402
from helpers import seeded_weights, load_case                # noqa: E402
from test_step_gpu import build_model, cosine                # noqa: E402


def _oracle_autocast_losses(cfg, params, batches, steps, warmup, t_total, seed):
    """The yardstick: the same loop on the CPU oracle with forward / backward under torch's bf16 autocast (the reference's own
    mixed-precision mode, train.py:172) and the fp32 AdamW of the oracle -- what bf16 alone does to this trajectory."""
    cpu_params = dict(params, device=torch.device("cpu"))
    sd = seeded_weights(cfg, cpu_params, base_seed=seed)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    losses = []
    for it in range(steps):
        for p in sd.values():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = O.oracle_step(sd, cfg, cpu_params, batches[it % len(batches)], cls_dropout=0.0)
        out[0].float().backward()
        losses.append(float(out[0]))
        lr = O.scheduled_lr(params["lr"], it, warmup, t_total, params["min_lr"])
        with torch.no_grad():
            for k, p in sd.items():
                if p.grad is None:
                    continue
                wd = 0.0 if any(nd in ("bert_pretrained." + k) for nd in O.NO_DECAY) else params["wd"]
                O.adamw_reference_step(p, p.grad.float(), m[k], v2[k], it + 1, lr, wd)
    return losses


def _run_both(cfg, params, batches, steps, warmup, t_total, seed):
    model, params = build_model(cfg, params, weights=None, seed=seed)
    core = model.bert_pretrained
    opt = get_optimizer(params, model)
    sched = WarmupLinearScheduleNonZero(opt, warmup_steps=warmup, t_total=t_total, min_lr=params["min_lr"])
    cpu_params = dict(params, device=torch.device("cpu"))
    sd = seeded_weights(cfg, cpu_params, base_seed=seed)
    init = {k: v.detach().clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    hip, ref = [], []
    for it in range(steps):
        batch = batches[it % len(batches)]
        # ---- HIP: train.py:173, 208-215 without the scaler
        loss = step_forward(model, batch, params)[0]
        loss.backward()
        opt.step()
        opt.zero_grad()
        sched.step()
        hip.append(loss.detach())
        # ---- oracle: the same step in fp32 on the CPU
        for p in sd.values():
            p.grad = None
        out = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
        out[0].backward()
        ref.append(float(out[0]))
        lr = O.scheduled_lr(params["lr"], it, warmup, t_total, params["min_lr"])
        with torch.no_grad():
            for k, p in sd.items():
                if p.grad is None:
                    continue
                wd = 0.0 if any(nd in ("bert_pretrained." + k) for nd in O.NO_DECAY) else params["wd"]
                O.adamw_reference_step(p, p.grad, m[k], v2[k], it + 1, lr, wd)
    opt.synchronize()
    torch.cuda.synchronize()
    hip = [float(x) for x in hip]
    named = dict(core.named_parameters())
    p_cos, u_cos = [], []
    for k, p in sd.items():
        got = named[k].detach().float().cpu()
        du_ref = p.detach() - init[k]
        if float(du_ref.norm()) == 0.0:                        # never-used tensors: untouched on both sides
            assert torch.equal(got, init[k]), k
            continue
        p_cos.append((cosine(got, p.detach()), k))
        if k.endswith(("key.bias", "key1.bias", "key2.bias")):   # mathematically zero gradient (softmax shift invariance): their update is
            continue                                             # Adam's normalisation of rounding noise, on both sides
        u_cos.append((cosine(got - init[k], du_ref), k))
    return hip, ref, sorted(p_cos), sorted(u_cos)


def test_tiny_model_tracks_the_oracle_over_forty_adamw_steps():
    torch.set_num_threads(8)
    cfg = C.tiny_config()
    params = C.default_params(categories=9)
    batches = [S.make_batch(3, 7, 5, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=40 + i) for i in range(4)]
    hip, ref, p_cos, u_cos = _run_both(cfg, params, batches, steps=40, warmup=8, t_total=60, seed=7)
    worst = max(abs(a - b) / abs(b) for a, b in zip(hip, ref))
    med, p10 = u_cos[len(u_cos) // 2][0], u_cos[len(u_cos) // 10][0]
    print("tiny trajectory: loss %.5f -> %.5f (oracle %.5f -> %.5f), worst pointwise %.2e; parameter cosine min %.6f; update cosine min %.4f p10 %.4f median %.4f"
          % (hip[0], hip[-1], ref[0], ref[-1], worst, p_cos[0][0], u_cos[0][0], p10, med), u_cos[:3])
    assert worst <= 2e-2, (worst, list(zip(hip, ref))[:5])
    assert p_cos[0][0] >= 0.999, p_cos[:5]
    assert med >= 0.97 and p10 >= 0.90, (med, p10, u_cos[:8])
    assert ref[-1] < ref[0] and hip[-1] < hip[0]                # the pool is being learnt on both sides


def test_full_depth_model_tracks_the_oracle_over_five_adamw_steps():
    """vilbert.json, 5 AdamW steps on ONE batch each of three draws: the committed B = 4 fixture's inputs (full_B4_V36_T20_F1024:
    reference-generated), another B = 4 batch and a B = 16 batch.

    What can be asked of a loss curve here: Adam's first steps move all 252 M weights by +- lr along the SIGN of gradients whose small
    entries are rounding noise in any bf16 pipeline, and a 4-row batch is overfitted within five steps (loss 0.71 -> 0.43), so the
    curve of step 3 onward is one draw of that noise: over five draws this path deviates from the fp32 oracle by 0.2 - 3.8 % at its
    worst step and the fp32 oracle's own bf16-autocast self by 0.3 - 2.9 %, on DIFFERENT draws (profiles/r6_trajectory_draws.txt).
    That was measured with the residual stream stored as bf16.  With it in fp32 (CrctStepCfg.residual_fp32, the default since the end of
    round 6) this path's worst step over the three draws below reads 0.59 % / 0.41 % / 0.16 % (the autocast oracle 0.41 % / 2.59 % / 0.34 %):
    the 2 % pointwise bound VERDICT r5 asked for holds.  Asserted: the first two steps within 1 %; EVERY step of every draw within 2 %; the
    mean over the draws of the worst deviation within 1.5 x the bf16-autocast oracle's, computed here on the same draws; final parameters
    at cosine >= 0.999 per tensor and the 5-step UPDATE of every tensor at median >= 0.90 / 10th percentile >= 0.75 (fixture draw)."""
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 1))
    z, meta, cfg, params, batch = load_case("full_B4_V36_T20_F1024")
    draws = [("fixture B4", batch), ("B4 seed 1", S.make_batch(4, 20, 36, 1024, seed=1)), ("B16 seed 4", S.make_batch(16, 20, 36, 1024, seed=4))]
    worst_hip, worst_yard = [], []
    for name, b in draws:
        hip, ref, p_cos, u_cos = _run_both(cfg, params, [b], steps=5, warmup=2, t_total=10, seed=meta["weight_seed"])
        yard = _oracle_autocast_losses(cfg, params, [b], 5, 2, 10, meta["weight_seed"])
        dev = [abs(a - r) / abs(r) for a, r in zip(hip, ref)]
        dev_y = [abs(a - r) / abs(r) for a, r in zip(yard, ref)]
        med, p10 = u_cos[len(u_cos) // 2][0], u_cos[len(u_cos) // 10][0]
        print("full-depth trajectory [%s]: loss %s (oracle %s; bf16-autocast oracle %s), worst pointwise %.2e (autocast %.2e); parameter cosine min %.6f; "
              "update cosine min %.4f p10 %.4f median %.4f" % (name, np.round(hip, 5).tolist(), np.round(ref, 5).tolist(), np.round(yard, 5).tolist(), max(dev),
                                                               max(dev_y), p_cos[0][0], u_cos[0][0], p10, med))
        if name.startswith("fixture"):
            assert abs(ref[0] - float(z["out.loss"])) < 1e-5            # the oracle starts on the reference's own number
            assert med >= 0.90 and p10 >= 0.75, (med, p10, u_cos[:8])
        assert max(dev[:2]) <= 1e-2 and max(dev) <= 2e-2, (name, dev)
        assert p_cos[0][0] >= 0.999, p_cos[:5]
        assert ref[-1] < ref[0] and hip[-1] < hip[0]
        worst_hip.append(max(dev))
        worst_yard.append(max(dev_y))
    print("worst-step deviation per draw: this path %s, bf16-autocast oracle %s" % (np.round(worst_hip, 4).tolist(), np.round(worst_yard, 4).tolist()))
    assert np.mean(worst_hip) <= 1.5 * np.mean(worst_yard), (worst_hip, worst_yard)
