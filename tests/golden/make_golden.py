"""Generate the golden fixtures by RUNNING THE REFERENCE (build container only).

Imports levymsn/CQA-CRCT from /root/reference on CPU (three sys.modules shims),
builds ``BertForMultiModalPreTraining`` the way ``VisualDialogEncoder.__init__`` /
``from_pretrained`` do before weight loading (encoder_decorator.py:11-17, vilbert.py:1205), fills
it with name-keyed seeded weights, feeds seeded synthetic batches through the reference's own
``encoder_decorator.forward`` + ``loss.backward()``, and writes inputs / outputs / gradients as
.npz files next to this script.  Reference source never enters the repo; only these vectors do.

    python tests/golden/make_golden.py            # regenerates every fixture
    python tests/golden/make_golden.py long       # only the long-sequence fixtures (T = 124 / 130)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "cqa-crct_amd"))
sys.path.insert(0, ROOT)
REF = "/root/reference/CRCT"

from crct import config as C          # noqa: E402
from crct import synthetic as S       # noqa: E402
from oracle import crct_oracle as O   # noqa: E402


def import_reference():
    m = types.ModuleType("pytorch_pretrained_bert")
    fu = types.ModuleType("pytorch_pretrained_bert.file_utils")
    fu.cached_path = lambda *a, **k: (_ for _ in ()).throw(EnvironmentError("no network"))
    m.file_utils = fu
    sys.modules["pytorch_pretrained_bert"] = m
    sys.modules["pytorch_pretrained_bert.file_utils"] = fu
    sys.path.insert(0, REF)
    from backbone import vilbert, encoder_decorator   # type: ignore
    return vilbert, encoder_decorator


def build_reference_model(vilbert, encoder_decorator, cfg, params):
    rcfg = vilbert.BertConfig.from_dict(cfg.to_dict())
    model = encoder_decorator.VisualDialogEncoder.__new__(encoder_decorator.VisualDialogEncoder)
    torch.nn.Module.__init__(model)
    model.bert_pretrained = vilbert.BertForMultiModalPreTraining(rcfg, params=params)
    model.bert_pretrained.cls.dropout.p = 0.0      # hard-coded 0.1 at vilbert.py:1045
    model.train()
    return model


def run_case(name, cfg, params, batch, vilbert, ed, save_weights, evaluation=False, grad_sample=None, yardstick=False):
    params = dict(params)
    params["device"] = torch.device("cpu")
    model = build_reference_model(vilbert, ed, cfg, params)
    S.seeded_fill_(model.state_dict(), base_seed=7)
    sd = {k[len("bert_pretrained."):]: v for k, v in model.named_parameters()}
    # -------- reference
    out = ed.forward(model, {k: v.clone() for k, v in batch.items()}, params, evaluation=evaluation)
    if evaluation:
        loss, lm, nsp, img, scores, reg = out
    else:
        loss, lm, nsp, img, scores, reg, leg = out
        loss.backward()
    # -------- oracle on the same weights (sanity; the test suite re-checks from the files)
    osd = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    oout = O.oracle_step(osd, cfg, params, batch, evaluation=evaluation, training=True, taps=taps,
                         cls_dropout=0.0)
    if not evaluation:
        oout[0].backward()
        d = abs(float(oout[0]) - float(loss))
        print("  [%s] loss ref %.8f oracle %.8f |d|=%.2e" % (name, float(loss), float(oout[0]), d))
        assert d < 1e-5
        worst = 0.0
        for k, p in sd.items():
            if p.grad is None:
                assert osd[k].grad is None or float(osd[k].grad.abs().max()) == 0.0, k
                continue
            # key biases have a mathematically zero gradient (softmax shift invariance): 1e-10 noise
            den = float(p.grad.abs().max()) + 1e-6
            worst = max(worst, float((p.grad - osd[k].grad).abs().max()) / den)
        print("  [%s] worst relative grad diff %.2e" % (name, worst))
        assert worst < 1e-3
    assert float((oout[4] - scores).abs().max()) < 1e-5
    assert float((oout[5][0] - reg[0]).abs().max()) < 1e-3

    rec = {"in." + k: v.numpy() for k, v in batch.items()}
    rec["out.nsp_scores"] = scores.detach().numpy()
    rec["out.reg_pred"] = reg[0].detach().numpy()
    rec["out.reg_loss"] = reg[1].detach().numpy()
    rec["out.reg_l1"] = reg[2].detach().numpy()
    rec["out.reg_right"] = np.array(reg[3], dtype=np.int64)
    rec["out.reg_dist5"] = reg[4].detach().numpy()
    # hidden states of the reference at the heads' inputs (recomputed through the reference modules)
    with torch.no_grad():
        bp = model.bert_pretrained
        T = batch["tokens"].shape[1]
        key_t = O.text_key_mask(batch["sep_indices"], batch["hist_len"], T)
        seq_t, seq_v, _, _, _ = bp.bert(batch["tokens"], batch["loc"], batch["image_feat"], batch["image_loc"],
                                        token_type_ids=batch["segments"], attention_mask=key_t,
                                        image_attention_mask=batch["image_mask"],
                                        image_target=batch["image_target"])
        rec["out.seq_t_cls"] = seq_t[:, 0].numpy()
        rec["out.seq_v_img"] = seq_v[:, 0].numpy()
        rec["out.emb_t"] = bp.bert.embeddings(batch["tokens"], token_type_ids=batch["segments"],
                                              loc=batch["loc"]).numpy()
        rec["out.emb_v"] = bp.bert.v_embeddings(batch["image_feat"], batch["image_loc"],
                                                batch["image_target"], None).numpy()
    if not evaluation:
        rec["out.loss"] = np.array(float(loss), dtype=np.float64)
        rec["out.nsp_loss"] = nsp.detach().numpy()
        for k, p in sd.items():
            if p.grad is None:
                rec["gradnorm." + k] = np.array(-1.0)          # never receives a gradient
                continue
            g = p.grad
            rec["gradnorm." + k] = np.array(float(g.double().norm()))
            if save_weights:
                rec["grad." + k] = g.numpy()
            else:
                flat = g.reshape(-1)
                n = min(flat.numel(), 64)
                idx = (torch.arange(n, dtype=torch.int64) * (flat.numel() - 1)) // max(n - 1, 1)
                rec["gradidx." + k] = idx.numpy()
                rec["gradsample." + k] = flat[idx].numpy()
    if yardstick and not evaluation:
        # what bf16 itself costs on this draw: the oracle under torch's CPU bf16 autocast (the reference's mixed-precision mode,
        # train.py:172) against its fp32 self -- per tensor the ratio of the gradient norms (tests add |ratio - 1| to their norm bound)
        sd16 = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = O.oracle_step(sd16, cfg, params, batch, training=True, cls_dropout=0.0)
        o16[0].float().backward()
        dev = []
        for k, p in sd.items():
            if p.grad is None or sd16[k].grad is None or float(p.grad.norm()) < 1e-7:
                continue
            r = float(sd16[k].grad.double().norm() / p.grad.double().norm())
            rec["yardratio." + k] = np.array(r)
            dev.append(abs(r - 1))
        print("  [%s] bf16-autocast yardstick: gradient norm off by up to %.1f %% (median %.2f %%)" % (name, 100 * max(dev), 100 * sorted(dev)[len(dev) // 2]))
    if save_weights:
        for k, p in sd.items():
            rec["w." + k] = p.detach().numpy()
    meta = dict(cfg=cfg.to_dict(), params={k: v for k, v in params.items() if k != "device"},
                evaluation=evaluation, weight_seed=7, cls_dropout=0.0)
    rec["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print("  wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))
    return model


def optimizer_case(vilbert, ed):
    """One AdamW + WarmupLinearScheduleNonZero step of the reference's own utils.py on the tiny model."""
    sys.modules.setdefault("pandas", __import__("pandas"))
    import importlib
    utils = importlib.import_module("utils")
    cfg = C.tiny_config()
    params = C.default_params(categories=9, L1=True, device=torch.device("cpu"))
    model = build_reference_model(vilbert, ed, cfg, params)
    S.seeded_fill_(model.state_dict(), base_seed=7)
    batch = S.make_batch(3, 7, 5, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=11)
    cwd = os.getcwd()
    os.chdir(REF)                      # get_optimizer opens config/language_weights.json relative to CWD
    try:
        opt = utils.get_optimizer(params, model)
    finally:
        os.chdir(cwd)
    sched = utils.WarmupLinearScheduleNonZero(opt, warmup_steps=4, t_total=10, min_lr=1.3e-5)
    rec = {}
    lrs = []
    for it in range(3):
        out = ed.forward(model, {k: v.clone() for k, v in batch.items()}, params)
        out[0].backward()
        opt.step()
        opt.zero_grad()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    for k, p in model.named_parameters():
        rec["w3." + k[len("bert_pretrained."):]] = p.detach().numpy()
    rec["lrs"] = np.array(lrs)
    rec["groups_lr_wd"] = np.array([[g["lr"], g["weight_decay"]] for g in opt.param_groups])
    # schedule table (utils.py:22-29)
    steps = [0, 1, 2999, 3000, 3001, 50000, 59999, 60000, 70000]
    s2 = utils.WarmupLinearScheduleNonZero(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=2e-5),
                                           warmup_steps=3000, t_total=60000, min_lr=1.3e-5)
    vals = []
    for st in steps:
        s2.last_epoch = st
        vals.append(s2.get_lr()[0])
    rec["sched_steps"] = np.array(steps)
    rec["sched_lr"] = np.array(vals)
    path = os.path.join(HERE, "tiny_adamw3.npz")
    np.savez_compressed(path, **rec)
    print("  wrote", path)


def checkpoint_schema_case(vilbert, ed):
    """Structure of the checkpoint the reference's train.py:284-291 writes (keys, shapes, dtypes, optimizer and
    scheduler state layout), taken from a real torch.save / torch.load round trip of the reference objects after
    two optimizer steps on the tiny model.  Only the structure is committed (values are covered by tiny_adamw3)."""
    import importlib
    import tempfile
    utils = importlib.import_module("utils")
    cfg = C.tiny_config()
    params = C.default_params(categories=9, L1=True, device=torch.device("cpu"))
    model = build_reference_model(vilbert, ed, cfg, params)
    S.seeded_fill_(model.state_dict(), base_seed=7)
    batch = S.make_batch(3, 7, 5, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=11)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        opt = utils.get_optimizer(params, model)
    finally:
        os.chdir(cwd)
    sched = utils.WarmupLinearScheduleNonZero(opt, warmup_steps=4, t_total=10, min_lr=1.3e-5)
    for it in range(2):
        ed.forward(model, {k: v.clone() for k, v in batch.items()}, params)[0].backward()
        opt.step()
        opt.zero_grad()
        sched.step()
    ckpt = {"model_state_dict": model.state_dict(), "scheduler_state_dict": sched.state_dict(),
            "optimizer_state_dict": opt.state_dict(), "iter_id": 2}                 # train.py:287-289
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "plotqa_encoder_0_2.ckpt")                           # train.py:282
        torch.save(ckpt, path)
        back = torch.load(path, map_location="cpu", weights_only=False)

    def jsonable(v):
        if isinstance(v, torch.Tensor):
            return {"tensor": list(v.shape), "dtype": str(v.dtype)}
        if isinstance(v, (list, tuple)):
            return [jsonable(x) for x in v]
        if isinstance(v, dict):
            return {str(k): jsonable(x) for k, x in v.items()}
        if isinstance(v, (int, float, bool, str)) or v is None:
            return v
        return repr(v)

    osd = back["optimizer_state_dict"]
    schema = {
        "file_name_pattern": "plotqa_encoder_%d_%d.ckpt",
        "top_level_keys": list(back.keys()),
        "iter_id": back["iter_id"],
        "model_state_dict": [[k, list(v.shape), str(v.dtype)] for k, v in back["model_state_dict"].items()],
        "optimizer_state_keys": sorted(osd.keys()),
        "optimizer_param_group_keys": sorted(osd["param_groups"][0].keys()),
        "optimizer_param_groups": [[g["lr"], g["weight_decay"], g["params"], list(g["betas"]), g["eps"], g.get("initial_lr")]
                                   for g in osd["param_groups"]],
        "optimizer_state_ids": sorted(int(k) for k in osd["state"].keys()),
        "optimizer_state_entry": jsonable(osd["state"][sorted(osd["state"].keys())[0]]),
        "scheduler_state_dict": jsonable(back["scheduler_state_dict"]),
    }
    path = os.path.join(HERE, "ckpt_schema.json")
    with open(path, "w") as f:
        json.dump(schema, f)
    print("  wrote %s (%.1f KB); %d model keys, %d groups, %d state entries" % (
        path, os.path.getsize(path) / 1024, len(schema["model_state_dict"]), len(osd["param_groups"]), len(osd["state"])))


def make_eval_batches(cfg, n_batches=3, seed=5):
    """Synthetic evaluation batches in the schema AFTER fig_dataloader.cut_batch_padding (:697-702): per-candidate rows
    concatenated over the questions, per-question tensors [Q, 1], string lists qid / qa_type."""
    g = np.random.RandomState(seed)
    qids = ["S3", "S17", "D6", "D14", "D2", "A2", "M1", "CD7", "D15", "D16", "C5", "S0"]
    types = ["line", "vbar", "hbar", "dot"]
    batches, ans_type = [], {}
    qa = 1000
    for bi in range(n_batches):
        Q = 4 + bi
        num_ans = g.randint(1, 7, size=Q)
        N = int(num_ans.sum())
        rows = S.make_batch(N, 7, 5, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=300 + bi)
        needs = g.rand(Q) < 0.5
        tol = np.where(needs, g.choice([0.01, 0.05, 0.2], size=Q), 0.0).astype(np.float32)
        scale = np.where(needs, g.uniform(0.5, 2.0, size=Q), 0.0).astype(np.float32)
        gt = np.where(needs, g.uniform(-1.5, 1.5, size=Q) * scale, 0.0).astype(np.float32)
        R = np.zeros((N, 4), dtype=np.float32)
        off = 0
        for q in range(Q):
            R[off:off + num_ans[q]] = [gt[q], 1.0 if needs[q] else 0.0, tol[q], scale[q]]
            off += num_ans[q]
        rows["R"] = torch.from_numpy(R)
        b = dict(rows)
        b["num_ans"] = torch.from_numpy(num_ans.astype(np.int64)).view(Q, 1)
        b["id"] = torch.arange(qa, qa + Q, dtype=torch.int64).view(Q, 1)
        b["gt_id"] = torch.from_numpy(np.array([g.randint(0, n) for n in num_ans], dtype=np.int64)).view(Q, 1)
        b["needs_reg"] = torch.from_numpy(needs).view(Q, 1)
        b["tolerance_margin"] = torch.from_numpy(tol).view(Q, 1)
        b["gt"] = torch.from_numpy(gt).view(Q, 1)
        b["reg_target"] = torch.from_numpy(np.where(needs, gt / np.where(scale == 0, 1, scale), 0).astype(np.float32)).view(Q, 1)
        b["qid"] = [qids[g.randint(len(qids))] for _ in range(Q)]
        b["qa_type"] = [types[g.randint(4)] for _ in range(Q)]
        for q in range(Q):
            ans_type[qa + q] = 2 if needs[q] else int(g.randint(0, 3))     # regression questions are answer kind 2 (:483)
        qa += Q
        batches.append(b)
    return batches, ans_type


def eval_scoring_case(vilbert, ed):
    """The reference's own evaluation loop (evaluation.py:199-386) on synthetic candidate batches, single-process gloo
    group, tensors kept on the CPU (``.cuda()`` is the identity here).  Commits inputs, the per-row forward outputs and
    the accuracy tables it returned."""
    import torch.distributed as dist
    pt = types.ModuleType("pytorch_transformers")
    tb = types.ModuleType("pytorch_transformers.tokenization_bert")
    tb.BertTokenizer = type("BertTokenizer", (), {})
    pt.tokenization_bert = tb
    sys.modules["pytorch_transformers"] = pt
    sys.modules["pytorch_transformers.tokenization_bert"] = tb
    sys.modules.setdefault("pandas", __import__("pandas"))
    import importlib
    evaluation = importlib.import_module("evaluation")
    cfg = C.tiny_config()
    params = C.default_params(categories=9, L1=True, device=torch.device("cpu"), ddp=True, world_size=1, rank=0,
                              save_path="/tmp", eval_set="val", start_checkpoint="none/tiny.ckpt", dataset="plotqa")
    model = build_reference_model(vilbert, ed, cfg, params)
    S.seeded_fill_(model.state_dict(), base_seed=7)
    batches, ans_type = make_eval_batches(cfg)
    # labels designed from the model's own predictions so that every branch of the scoring is hit: classification right /
    # wrong, regression within 5 % and within the tick tolerance, within one of the two only, wrong candidate chosen
    model.eval()
    case = 0
    for b in batches:
        Q = b["num_ans"].shape[0]
        probe = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
        probe["R"] = torch.tensor([[1.0, 1.0, 0.0, 1.0]]).repeat(b["tokens"].shape[0], 1)     # needs=1, scale=1: reg[0] = raw r
        with torch.no_grad():
            out = ed.forward(model, probe, params, output_nsp_scores=True, evaluation=True)
        p0 = torch.softmax(out[4], dim=1)[:, 0]
        r_raw = out[5][0]
        off = 0
        for q in range(Q):
            n = int(b["num_ans"][q])
            a = int(torch.argmax(p0[off:off + n]))
            r = float(r_raw[off + a])
            kind = case % 6
            case += 1
            needs, gt_id, tol, scale, target = False, a, 0.0, 0.0, 0.0
            if kind == 1:
                gt_id = (a + 1) % n if n > 1 else a
            elif kind == 2:
                needs, scale, target, tol = True, 2.0, r * 1.02, 0.5                 # both measures right
            elif kind == 3:
                needs, scale, target, tol = True, 0.5, r * 1.03, abs(r) * 0.001      # 5 % right, tick tolerance wrong
            elif kind == 4:
                needs, scale, target, tol = True, 1.5, r * 1.5, abs(r) + 1.0         # 5 % wrong, tick tolerance right
            elif kind == 5:
                needs, scale, target, tol = True, 1.0, r * 1.01, 0.5
                gt_id = (a + 1) % n if n > 1 else a                                  # regression fine, wrong candidate
            b["R"][off:off + n] = torch.tensor([target * scale, 1.0 if needs else 0.0, tol, scale])
            b["gt_id"][q, 0] = gt_id
            b["needs_reg"][q, 0] = needs
            b["tolerance_margin"][q, 0] = tol
            b["gt"][q, 0] = target * scale
            b["reg_target"][q, 0] = target
            ans_type[int(b["id"][q])] = 2 if needs else (int(b["id"][q]) % 2)
            off += n
    model.train()

    class Dataset(object):                      # the two dataset hooks the loop calls (fig_dataloader.py:697-715)
        def cut_batch_padding(self, item):
            pass                                # the synthetic batches are already cut

        def get_ans_type(self, qa_ind):
            return ans_type[int(qa_ind)]

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
    old_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        total, breakdown = evaluation.plotqa_evaluate_DDP([{k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in b.items()} for b in batches],
                                                          Dataset(), params, 4, model, progress=False, csv=False)
        hist = torch.zeros(13).long()
        dist_vals = torch.tensor([0.0, 0.01, 0.05, 0.0500001, 0.1, 0.12, 0.2, 0.25, 0.3, 0.55, 0.9, 1.0, 1.00001, 7.0, 0.95])
        evaluation.reduce_histogram(hist, dist_vals, dist.new_group([0]))
    finally:
        torch.Tensor.cuda = old_cuda
        dist.destroy_process_group()
    rec = {"total_correct": total.numpy(), "breakdown": breakdown.numpy(), "hist_in": dist_vals.numpy(), "hist_out": hist.numpy(),
           "n_batches": np.array(len(batches)), "eval_batch_size": np.array(4)}
    model.eval()
    for bi, b in enumerate(batches):
        with torch.no_grad():
            out = ed.forward(model, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}, params,
                             output_nsp_scores=True, evaluation=True)
        rec["b%d.nsp_scores" % bi] = out[4].numpy()
        for j in (0, 2, 4):
            rec["b%d.reg%d" % (bi, j)] = out[5][j].numpy()
        for k, v in b.items():
            if torch.is_tensor(v):
                rec["b%d.in.%s" % (bi, k)] = v.numpy()
        rec["b%d.qid" % bi] = np.array(b["qid"])
        rec["b%d.qa_type" % bi] = np.array(b["qa_type"])
        rec["b%d.ans_type" % bi] = np.array([ans_type[int(i)] for i in b["id"].view(-1)])
    rec["meta"] = np.array(json.dumps(dict(cfg=cfg.to_dict(), params={k: v for k, v in params.items() if k != "device"},
                                           weight_seed=7)))
    path = os.path.join(HERE, "tiny_evalscore.npz")
    np.savez_compressed(path, **rec)
    print("  wrote %s (%.1f KB); total_correct =\n%s" % (path, os.path.getsize(path) / 1024, total.numpy()))


def small_long_config():
    """A small model with the head sizes the long-sequence attention kernels take
    (32 / 48 / 32) and a position table that covers 130 tokens."""
    return C.tiny_config(hidden_size=128, num_attention_heads=4, intermediate_size=256, max_position_embeddings=160,
                         v_hidden_size=192, v_num_attention_heads=4, v_intermediate_size=192, bi_hidden_size=128,
                         bi_num_attention_heads=4)


def long_sequence_cases(vilbert, ed):
    """Sequences beyond 112 tokens: the reference's own PlotQA shape (config/plotqa.json:5-6 max_vis_features 44, max_seq_len 124;
    v_feature_size 1024 of config/vilbert.json) at full depth with the padding real samples carry (utils.py:152-160), and a
    small model at T = 130 / V = 9 with padding in both streams (name-keyed seeded weights; outputs, activations at the heads,
    gradient norms and samples of every tensor committed -- the oracle, asserted here to match the reference's full gradients
    to 1e-3 of each tensor's maximum, supplies full tensors at test time)."""
    cfg = C.vilbert_config(v_feature_size=1024, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                           v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
    # seed 1238: a signal-rich draw (coherence of the per-row gradient sum at the CLS rows 0.44; the oracle under bf16 autocast keeps
    # min / median cosine 0.988 / 0.993 against its fp32 self).  Seed 1234 at this shape is a cancelling draw with 2 + 2 labels on 4 rows:
    # a 0.003 error of a logit (tolerance 0.03) moves the visual pooler's gradient norm by 20 % (profiles/r6_long_attn_parity.txt) --
    # a fixture with sample cosines and 5 % norm bounds needs a draw whose gradients are not the residue of a cancellation.
    batch = S.make_batch(4, 124, 44, 1024, seed=1238, lengths=[124, 71, 96, 110], n_vis=[44, 29, 37, 44])
    batch["R"][:, 1] = torch.tensor([1.0, 1.0, 0.0, 1.0])      # three regression rows: the regressor's gradients are not one row's outer product
    batch["needs_reg"] = (batch["R"][:, 1:2] == 1)
    batch["image_feat"] = batch["image_feat"].half().float()
    run_case("full_B4_V44_T124_F1024", cfg, C.default_params(), batch, vilbert, ed, save_weights=False, yardstick=True)
    small = small_long_config()
    b = S.make_batch(3, 130, 9, small.v_feature_size, categories=9, vocab_size=small.vocab_size, seed=21,
                     lengths=[130, 97, 113], n_vis=[9, 4, 7])
    b["R"][0, 1] = 1.0
    b["R"][1, 1] = 0.0
    b["R"][2, 1] = 1.0
    b["needs_reg"] = (b["R"][:, 1:2] == 1)
    run_case("small_B3_V9_T130", small, C.default_params(categories=9), b, vilbert, ed, save_weights=False, yardstick=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vilbert, ed = import_reference()
    # ---- tiny config, full weights + all gradients committed
    tiny = C.tiny_config()
    p_tiny = C.default_params(categories=9)
    b = S.make_batch(3, 7, 5, tiny.v_feature_size, categories=9, vocab_size=tiny.vocab_size, seed=11)
    # edge cases for the embeddings: all-zero loc rows, segments incl. -1,0,1,2..11
    b["segments"][0, 1] = 2
    b["segments"][1, 1] = 3
    b["loc"][0, 1] = 0
    b["R"][0, 1] = 1.0
    b["R"][1, 1] = 0.0
    b["R"][2, 1] = 1.0
    b["R"][2, 0] = 0.0                      # target == 0 rule (vilbert.py:1633)
    b["next_sentence_labels"][1, 0] = -1    # ignore_index row
    b["needs_reg"] = (b["R"][:, 1:2] == 1)
    run_case("tiny_L1", tiny, p_tiny, b, vilbert, ed, save_weights=True)
    p_s = dict(p_tiny, L1=False)
    b2 = {k: v.clone() for k, v in b.items()}
    b2["R"][0, 0] = 250.0                   # |target| > 1 -> zeroed in 'L1_smooth' kind (vilbert.py:1641)
    run_case("tiny_smoothL1", tiny, p_s, b2, vilbert, ed, save_weights=False)
    run_case("tiny_eval", tiny, p_tiny, b2, vilbert, ed, save_weights=False, evaluation=True)
    optimizer_case(vilbert, ed)
    checkpoint_schema_case(vilbert, ed)
    eval_scoring_case(vilbert, ed)
    # ---- full vilbert.json shapes, seeded weights: only inputs / outputs / gradient samples committed
    for nm, B, V, T, Fv in (("full_B4_V36_T20_F1024", 4, 36, 20, 1024),
                            ("full_B4_V36_T20_F2048", 4, 36, 20, 2048),
                            ("full_B2_V100_T40_F2048", 2, 100, 40, 2048)):
        cfg = C.vilbert_config(v_feature_size=Fv, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                               v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
        params = C.default_params()
        batch = S.make_batch(B, T, V, Fv, seed=1234)
        # keep the committed inputs small: features rounded to fp16-representable values
        batch["image_feat"] = batch["image_feat"].half().float()
        run_case(nm, cfg, params, batch, vilbert, ed, save_weights=False)
    long_sequence_cases(vilbert, ed)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ckpt_schema":      # regenerate only the checkpoint schema
        sys.modules.setdefault("pandas", __import__("pandas"))
        checkpoint_schema_case(*import_reference())
    elif len(sys.argv) > 1 and sys.argv[1] == "evalscore":      # regenerate only the evaluation-scoring fixture
        eval_scoring_case(*import_reference())
    elif len(sys.argv) > 1 and sys.argv[1] == "long":           # regenerate only the long-sequence fixtures (round 6)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        long_sequence_cases(*import_reference())
    else:
        main()
