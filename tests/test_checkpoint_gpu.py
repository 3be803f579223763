"""Checkpoint save / resume against the reference's own checkpoint format.

tests/golden/ckpt_schema.json is the structure of a real checkpoint written by the reference's train.py code path
(torch.save of model / AdamW / scheduler state after two steps on the tiny model, made by tests/golden/make_golden.py);
tests/golden/tiny_adamw3.npz holds the reference's weights after three uninterrupted steps.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crct import config as C                       # noqa: E402
from crct import synthetic as S                    # noqa: E402
from crct import checkpoint as CK                  # noqa: E402
from crct.optim import get_optimizer, WarmupLinearScheduleNonZero   # noqa: E402
from crct.step_adapter import forward as step_forward   # noqa: E402
from helpers import GOLDEN                         # noqa: E402
from test_step_gpu import build_model              # noqa: E402


def _setup():
    zw = np.load(os.path.join(GOLDEN, "tiny_L1.npz"))
    cfg = C.tiny_config()
    base = C.default_params(categories=9, L1=True, warmup=4, min_lr=1.3e-5)
    batch = S.make_batch(3, 7, 5, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=11)
    model, params = build_model(cfg, base, weights=zw)
    opt = get_optimizer(params, model)
    return cfg, base, batch, model, params, opt


def _steps(model, params, opt, sched, batch, n):
    for _ in range(n):
        step_forward(model, batch, params)[0].backward()
        opt.step()
        opt.zero_grad()
        sched.step()


def test_saved_checkpoint_has_the_reference_structure(tmp_path):
    schema = json.load(open(os.path.join(GOLDEN, "ckpt_schema.json")))
    cfg, base, batch, model, params, opt = _setup()
    sched = WarmupLinearScheduleNonZero(opt, warmup_steps=4, t_total=10, min_lr=1.3e-5)
    _steps(model, params, opt, sched, batch, 2)
    path = CK.save_checkpoint(str(tmp_path), model, opt, sched, epoch=0, step_iter_id=1)
    assert os.path.basename(path) == schema["file_name_pattern"] % (0, 2)
    back = torch.load(path, map_location="cpu", weights_only=False)
    assert list(back.keys()) == schema["top_level_keys"] and back["iter_id"] == schema["iter_id"]
    got = [[k, list(v.shape), str(v.dtype)] for k, v in back["model_state_dict"].items()]
    assert got == schema["model_state_dict"]                                  # same keys, same ORDER, shapes, dtypes
    # independent CPU tensors, not views of the flat device buffers
    for v in back["model_state_dict"].values():
        assert v.device.type == "cpu" and v.untyped_storage().nbytes() == v.numel() * 4
    osd = back["optimizer_state_dict"]
    assert sorted(osd.keys()) == schema["optimizer_state_keys"]
    assert sorted(osd["param_groups"][0].keys()) == schema["optimizer_param_group_keys"]
    groups = [[g["lr"], g["weight_decay"], g["params"], list(g["betas"]), g["eps"], g.get("initial_lr")] for g in osd["param_groups"]]
    assert len(groups) == len(schema["optimizer_param_groups"])
    for a, b in zip(groups, schema["optimizer_param_groups"]):
        assert a[2] == b[2] and np.allclose([a[0], a[1], a[4], a[5]], [b[0], b[1], b[4], b[5]], rtol=0, atol=1e-12) and a[3] == b[3]
    assert sorted(int(k) for k in osd["state"].keys()) == schema["optimizer_state_ids"]     # only tensors that get gradients
    first = osd["state"][schema["optimizer_state_ids"][0]]
    ent = schema["optimizer_state_entry"]
    assert sorted(first.keys()) == sorted(ent.keys())
    for k in ent:
        assert list(first[k].shape) == ent[k]["tensor"] and str(first[k].dtype) == ent[k]["dtype"]
    assert float(first["step"]) == 2.0
    ss, ref = back["scheduler_state_dict"], schema["scheduler_state_dict"]
    for k in ("warmup_steps", "t_total", "min_lr", "last_epoch", "_step_count"):
        assert ss[k] == ref[k], k
    assert np.allclose(ss["base_lrs"], ref["base_lrs"]) and np.allclose(ss["_last_lr"], ref["_last_lr"])


def test_resume_continues_the_reference_trajectory(tmp_path):
    """2 steps -> save -> fresh model / optimizer -> ``-continue`` load -> 3rd step == the reference's 3 uninterrupted
    steps (same bound as the optimizer test: Adam moves an element by at most ~lr per step) and == our own
    uninterrupted run bit for bit on the tensors without atomics."""
    za = np.load(os.path.join(GOLDEN, "tiny_adamw3.npz"))
    cfg, base, batch, model, params, opt = _setup()
    sched = WarmupLinearScheduleNonZero(opt, warmup_steps=4, t_total=10, min_lr=1.3e-5)
    _steps(model, params, opt, sched, batch, 2)
    path = CK.save_checkpoint(str(tmp_path), model, opt, sched, epoch=0, step_iter_id=1)
    _steps(model, params, opt, sched, batch, 1)                       # the uninterrupted run goes on
    torch.cuda.synchronize()
    straight = {k: p.detach().float().cpu().clone() for k, p in model.bert_pretrained.named_parameters()}

    atomic = ("word_embeddings", "position_embeddings", "plotqa_type_embeddings", "color_emb")
    for restore_lr in (False, True):
        cfg2, base2, _, model2, params2, opt2 = _setup()
        for p in model2.parameters():                                # make sure everything really comes from the file
            p.data.zero_()
        iters_per_epoch = 1                                          # t_total = iters_per_epoch * 20 in train.py; the saved value wins
        sched2, start_iter, cont_epoch, _ = CK.resume(model2, opt2, path, params2, iters_per_epoch, restore_lr=restore_lr)
        assert start_iter == 2 and cont_epoch == 1
        assert sched2.t_total == 10 and sched2.last_epoch == 2
        # the reference's resume leaves the optimizer one schedule tick ahead (constructor step): lr(3) instead of lr(2)
        want = float(za["lrs"][1]) if restore_lr else float(za["lrs"][2])
        assert abs(opt2.param_groups[0]["lr"] - want) < 1e-12
        bound = 2 * (1.3e-5 * 2 + want) + 1e-7                        # Adam moves an element by at most ~lr per step
        _steps(model2, params2, opt2, sched2, batch, 1)
        opt2.synchronize()
        torch.cuda.synchronize()
        for k, p in model2.bert_pretrained.named_parameters():
            w = p.detach().float().cpu()
            assert float((w - torch.from_numpy(za["w3." + k])).abs().max()) <= bound, k
            if restore_lr and not any(a in k for a in atomic):
                assert torch.equal(w, straight[k]), k


def test_key_intersection_load_of_a_reference_style_file(tmp_path):
    """train.py:94-104 / evaluation.py:31-41: unknown keys are ignored, missing keys keep their initial value,
    a bare state_dict file works as well as the wrapped one."""
    zw = np.load(os.path.join(GOLDEN, "tiny_L1.npz"))
    cfg, base, batch, model, params, opt = _setup()
    names = [k for k, _ in model.state_dict().items()]
    sd = {k: torch.from_numpy(zw["w." + k[len("bert_pretrained."):].replace("cls.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight")]) + 1.0
          for k in names}
    dropped = names[5]
    del sd[dropped]
    sd["bert_pretrained.some.other.head.weight"] = torch.zeros(3)
    for wrapped in (True, False):
        path = os.path.join(str(tmp_path), "plotqa_encoder_3_77.ckpt")
        torch.save({"model_state_dict": sd, "iter_id": 77} if wrapped else sd, path)
        enc = CK.get_encoder(dict(params), ckpt=path, config=cfg)
        got = enc.state_dict()
        before = model.state_dict()
        for k in names:
            if k == dropped:
                continue
            assert torch.allclose(got[k].cpu(), sd[k]), k
        # the engine sees the loaded weights (bf16 shadow refreshed on the next forward)
        out = step_forward(enc, batch, dict(params))
        assert torch.isfinite(out[0])
    with pytest.raises(AssertionError):
        CK.load_model_weights(model, {"model_state_dict": {"nothing.known": torch.zeros(1)}})
