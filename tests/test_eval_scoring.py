"""Evaluation scoring path.

CPU: the oracle's scoring (oracle/eval_oracle.py) reproduces the accuracy tables the REFERENCE's own evaluation loop
returned for the committed candidate batches (tests/golden/tiny_evalscore.npz, made by tests/golden/make_golden.py
running CRCT/evaluation.py:plotqa_evaluate_DDP), and the oracle model reproduces its per-row forward outputs.
GPU: the HIP path (engine evaluation forwards + crct_eval_select + device-side tables, crct/evaluation.py) against the
oracle on identical inputs: selection / flags / tables are index and integer work -> exact.
"""
import json
import os

import numpy as np
import pytest
import torch

from crct import config as C
from helpers import GOLDEN
from oracle import crct_oracle as O
from oracle import eval_oracle as EO


def _load():
    z = np.load(os.path.join(GOLDEN, "tiny_evalscore.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    batches = []
    for bi in range(int(z["n_batches"])):
        b = {k[len("b%d.in." % bi):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("b%d.in." % bi)}
        b["qid"] = [str(x) for x in z["b%d.qid" % bi]]
        b["qa_type"] = [str(x) for x in z["b%d.qa_type" % bi]]
        batches.append(b)
    return z, meta, batches


def _oracle_tables(z, batches, rows_of):
    total = np.zeros((6, 2))
    breakdown = np.zeros((5, 4, 3, 3))
    picked = []
    for bi, b in enumerate(batches):
        nsp, r0, r2, r4 = rows_of(bi, b)
        p0 = EO.softmax_p0(nsp)
        ans, out, err, terr = EO.select_answers(p0, r0, r4, r2, b["num_ans"].numpy())
        nsp_right, reg_right, reg_t_right, correct, correct_t = EO.correctness(
            ans, err, terr, b["gt_id"].numpy(), b["needs_reg"].numpy(), b["tolerance_margin"].numpy())
        needs = b["needs_reg"].numpy().reshape(-1).astype(bool)
        total += EO.total_acc_increment(needs, nsp_right, reg_right, reg_t_right)
        breakdown += EO.breakdown_increment(b["qid"], b["qa_type"], z["b%d.ans_type" % bi], needs, correct, correct_t)
        picked.append((ans, out, err, terr))
    return total, breakdown, picked


def test_oracle_scoring_reproduces_the_reference_tables():
    z, meta, batches = _load()
    total, breakdown, _ = _oracle_tables(z, batches, lambda bi, b: (z["b%d.nsp_scores" % bi], z["b%d.reg0" % bi], z["b%d.reg2" % bi], z["b%d.reg4" % bi]))
    assert np.array_equal(total, z["total_correct"]), (total, z["total_correct"])
    assert np.array_equal(breakdown, z["breakdown"])
    assert 0 < total[0, 0] < total[0, 1] and 0 < total[2, 0] < total[2, 1] and total[4, 0] < total[0, 0]   # every branch populated
    assert np.array_equal(EO.histogram_increment(z["hist_in"]), z["hist_out"])
    # category rule (evaluation.py:437-449)
    assert [EO.qcat_of_qid(q) for q in ("S0", "S17", "S18", "D0", "D6", "D15", "D16", "A2", "CD7")] == [0, 0, 2, 1, 2, 1, 2, 2, 2]


def test_oracle_model_reproduces_the_reference_row_outputs():
    z, meta, batches = _load()
    cfg = C.BertConfig.from_dict(meta["cfg"])
    params = dict(meta["params"], device=torch.device("cpu"))
    from crct import synthetic as S
    from crct.layout import parameter_table
    table, _ = parameter_table(cfg, params)
    sd = {"bert_pretrained." + e.name: torch.zeros(e.shape) for e in table}
    S.seeded_fill_(sd, base_seed=meta["weight_seed"])
    sd = {k[len("bert_pretrained."):]: v for k, v in sd.items()}
    for bi, b in enumerate(batches):
        with torch.no_grad():
            out = O.oracle_step(sd, cfg, params, b, evaluation=True, training=False)
        assert np.allclose(out[4].numpy(), z["b%d.nsp_scores" % bi], atol=2e-6)
        for j in (0, 2, 4):
            assert np.allclose(out[5][j].numpy(), z["b%d.reg%d" % (bi, j)], rtol=1e-4, atol=2e-6), (bi, j)


@pytest.mark.gpu
def test_hip_scoring_path_matches_the_oracle_exactly():
    from crct import evaluation as EV
    from test_step_gpu import build_model
    z, meta, batches = _load()
    cfg = C.BertConfig.from_dict(meta["cfg"])
    params = dict(meta["params"])
    model, params = build_model(cfg, params, seed=meta["weight_seed"])
    params["ddp"] = False
    ans_type = {}
    for bi, b in enumerate(batches):
        for i, t in zip(b["id"].view(-1).tolist(), z["b%d.ans_type" % bi].tolist()):
            ans_type[i] = t

    class Dataset(object):
        def cut_batch_padding(self, item):
            pass

        def get_ans_type(self, qa_ind):
            return ans_type[int(qa_ind)]

    total, breakdown, hist = EV.plotqa_evaluate([dict(b) for b in batches], Dataset(), params, int(z["eval_batch_size"]), model)
    assert model.training                                                  # restored (evaluation.py:384)
    # the oracle scoring on the HIP model's OWN row scores: everything after the forward is index / integer work -> exact
    model.eval()
    rows = {}
    with torch.no_grad():
        for bi, b in enumerate(batches):
            s, o, e, t = EV.score_rows(model, dict(b), params, int(z["eval_batch_size"]))
            rows[bi] = (s.float().cpu().numpy(), o.cpu().numpy(), t.cpu().numpy(), e.cpu().numpy())
            ans, so, se, st, p0 = EV.select_answers(s, o, e, t, b["num_ans"])
            ref = EO.select_answers(EO.softmax_p0(rows[bi][0]), rows[bi][1], rows[bi][3], rows[bi][2], b["num_ans"].numpy())
            assert np.array_equal(ans.cpu().numpy(), ref[0])
            assert np.array_equal(so.cpu().numpy(), ref[1]) and np.array_equal(se.cpu().numpy(), ref[2]) and np.array_equal(st.cpu().numpy(), ref[3])
            assert np.allclose(p0.cpu().numpy(), EO.softmax_p0(rows[bi][0]), atol=1e-6)
            # forced answers ('_REGS' question files) and an out-of-range id
            forced = b["gt_id"].clone()
            fa, fo, fe, ft, _ = EV.select_answers(s, o, e, t, b["num_ans"], forced)
            rf = EO.select_answers(EO.softmax_p0(rows[bi][0]), rows[bi][1], rows[bi][3], rows[bi][2], b["num_ans"].numpy(), forced.view(-1).numpy())
            assert np.array_equal(fa.cpu().numpy(), rf[0]) and np.array_equal(fe.cpu().numpy(), rf[2])
    model.train()
    o_total, o_breakdown, _ = _oracle_tables(z, batches, lambda bi, b: rows[bi])
    assert np.array_equal(total.cpu().numpy(), o_total), (total, o_total)
    assert np.array_equal(breakdown.cpu().numpy(), o_breakdown)
    # and against the reference's tables: bf16 forwards may flip a near-tie, so allow a small number of differing questions
    ref_total = z["total_correct"]
    assert np.array_equal(total.cpu().numpy()[:, 1], ref_total[:, 1])      # counts are data properties
    assert np.abs(total.cpu().numpy()[:, 0] - ref_total[:, 0]).max() <= 2
    assert int(hist.sum()) <= int(ref_total[2, 1])                          # one histogram entry per regression question at most
