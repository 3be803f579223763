"""Evaluation scoring path on the MI355X step engine.

Mirror of the scoring half of CRCT/evaluation.py ``plotqa_evaluate_DDP`` (:199-386) and its helpers
(``reduce_total_acc`` :492-525, ``reduce_histogram`` :528-549, ``reduce_breakdown_table`` :465-489,
``get_qcat_by_qid`` :437-449): the evaluation batch holds, per question, ``num_ans`` candidate answers as consecutive
rows; the rows are scored by chunked evaluation forwards (dropout off, ``'L1'`` regression kind), every question picks
the candidate with the highest answer probability, and regression questions count as right when the chosen candidate's
value is within 5 % (or within the chart's tick tolerance).

What is different from the reference is only HOW: the reference walks the questions in a Python loop with one
``.item()`` host sync each (:281-292) and moves every intermediate to the CPU; here the selection is one HIP launch
(``crct_eval_select``, one wave per question) and all flags / tables stay on the device until the caller reads them.
Logging, CSV dumps and plots of the reference are not rebuilt.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import lib as L
from .step_adapter import forward as step_forward

# fig_dataloader.py:27-32: per-candidate tensors, padded to [Q, 120, ...] by the collate step
PADDING_TXT = ["tokens", "segments", "sep_indices", "mask", "next_sentence_labels", "hist_len", "loc", "legend_belonging_t"]
PADDING_VIS = ["image_feat", "image_loc", "image_mask", "image_target", "image_label", "legend_belonging_v", "R"]
FIG_INDEX = {"Total": 0, "line": 1, "vbar": 2, "hbar": 3, "dot": 4}                # evaluation.py:467


def cut_batch_padding(batch, keys=None):
    """fig_dataloader.py:697-702: keep the first ``num_ans[i]`` candidate rows of question i and concatenate."""
    n = batch["num_ans"].reshape(-1).tolist()
    for k in (keys if keys is not None else PADDING_VIS + PADDING_TXT):
        if k in batch:
            x = batch[k]
            batch[k] = torch.cat([x[i, :n[i], ...] for i in range(x.shape[0])], dim=0)
    return batch


def get_qcat_by_qid(qid):
    """evaluation.py:437-449: ('s', 0) structural S0..S17, ('d', 1) data retrieval D0..D15 except D6, ('r', 2) reasoning."""
    num = qid[1:]
    if qid[:1] == "S" and num.isdigit() and 0 <= int(num) <= 17:
        return "s", 0
    if qid[:1] == "D" and num.isdigit() and 0 <= int(num) <= 15 and int(num) != 6:
        return "d", 1
    return "r", 2


def score_rows(dialog_encoder, batch, params, eval_batch_size):
    """evaluation.py:233-266: chunked evaluation forwards over all candidate rows.
    Returns device tensors (nsp_scores [N, 2], reg_output [N], reg_err [N], reg_t_err [N])."""
    N = batch["tokens"].shape[0]
    scores, out, err, terr = [], [], [], []
    for j in range(int(np.ceil(N / eval_batch_size))):
        lo, hi = j * eval_batch_size, min((j + 1) * eval_batch_size, N)
        _, _, _, _, nsp_scores, regression = step_forward(dialog_encoder, batch, params, output_nsp_scores=True,
                                                          evaluation=True, sample_ids=np.arange(lo, hi))
        assert nsp_scores.shape[-1] == 2                                           # :252
        scores.append(nsp_scores)
        out.append(regression[0]); err.append(regression[4]); terr.append(regression[2])
    return torch.cat(scores, 0), torch.cat(out, 0), torch.cat(err, 0), torch.cat(terr, 0)


def select_answers(nsp_scores, reg_out, reg_err, reg_terr, num_ans, forced=None):
    """evaluation.py:281-302 as ONE launch.  Returns (answers [Q] int64, out [Q], err [Q], t_err [Q], prob0 [N])."""
    dev = nsp_scores.device
    if dev.type != "cuda":
        raise RuntimeError("select_answers runs on an MI355X only (no CPU fallback)")
    na = num_ans.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
    Q, N = na.numel(), nsp_scores.shape[0]
    f = lambda t: t.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
    logits = nsp_scores.to(torch.float32).contiguous()
    ro, re_, rt = f(reg_out), f(reg_err), f(reg_terr)
    fz = forced.reshape(-1).to(device=dev, dtype=torch.int64).contiguous() if forced is not None else None
    answers = torch.empty(Q, dtype=torch.int64, device=dev)
    so, se, st = (torch.empty(Q, device=dev) for _ in range(3))
    p0 = torch.empty(N, device=dev)
    L.check(L.load().crct_eval_select(L.ptr(logits), L.ptr(ro), L.ptr(re_), L.ptr(rt), L.ptr(na), L.ptr(fz), Q, N, L.ptr(p0),
                                      L.ptr(answers), L.ptr(so), L.ptr(se), L.ptr(st), L.current_stream()), "eval_select")
    return answers, so, se, st, p0


def correctness(answers, sel_err, sel_terr, batch):
    """evaluation.py:303-311 on the device."""
    dev = answers.device
    gt_id = batch["gt_id"].reshape(-1).to(dev)
    needs = batch["needs_reg"].reshape(-1).to(dev).bool()
    tol = batch["tolerance_margin"].reshape(-1).to(dev).float()
    nsp_right = answers == gt_id
    reg_right = (sel_err <= 0.05) & needs
    reg_t_right = (sel_terr <= tol) & needs
    return nsp_right, reg_right, reg_t_right, needs


def reduce_total_acc(total_correct_tensor, needs_regression, nsp_right, reg_right, reg_t_right, group=None):
    """evaluation.py:492-525: the [6, 2] (hits, count) table; all-reduced when a process group is up."""
    t = torch.zeros_like(total_correct_tensor)
    n = nsp_right.shape[0]
    not_needed = needs_regression.logical_not()
    t[0, 0], t[0, 1] = nsp_right.sum(), n
    t[1, 0], t[1, 1] = (nsp_right & needs_regression).sum(), needs_regression.sum()
    t[2, 0], t[2, 1] = reg_right.sum(), needs_regression.sum()
    t[3, 0], t[3, 1] = reg_t_right.sum(), needs_regression.sum()
    t[4, 0], t[4, 1] = (nsp_right & (not_needed | reg_right)).sum(), n
    t[5, 0], t[5, 1] = (nsp_right & (not_needed | reg_t_right)).sum(), n
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    total_correct_tensor += t
    return total_correct_tensor


def reduce_histogram(histogram, reg_5_dist, group=None):
    """evaluation.py:528-549: 13 bins of the relative regression error."""
    d = reg_5_dist
    h = torch.zeros_like(histogram)
    k = 0
    for i in range(4):
        h[k] = (((i / 20) < d) & (d <= ((i + 1) / 20))).sum()
        k += 1
    for i in range(2, 10):
        h[k] = (((i / 10) < d) & (d <= ((i + 1) / 10))).sum()
        k += 1
    h[k] = (1 < d).sum()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    histogram += h
    return histogram


def reduce_breakdown_table(get_ans_type, group, params, breakdown_tensor, batch, correct, correct_t, needs):
    """evaluation.py:465-489.  The string / dataset look-ups of the questions run on the host (they are host data);
    the table itself is built by one scatter-add on the device."""
    dev = breakdown_tensor.device
    ids = batch["id"].reshape(-1).tolist()
    qc = [get_qcat_by_qid(q)[1] for q in batch["qid"]]
    fig = [FIG_INDEX[t] for t in batch["qa_type"]]
    ans = [int(get_ans_type(i)) for i in ids]
    qc_t, fig_t, ans_t = (torch.tensor(v, dtype=torch.int64, device=dev) for v in (qc, fig, ans))
    t = torch.zeros_like(breakdown_tensor)
    one = torch.ones(len(ids), dtype=t.dtype, device=dev)
    c5, ct = correct.to(t.dtype), correct_t.to(t.dtype)
    nd = needs.to(t.dtype)
    last = torch.full_like(ans_t, breakdown_tensor.shape[1] - 1)
    for f in (torch.zeros_like(fig_t), fig_t):
        for col, val in ((0, c5), (1, ct), (2, one)):
            c = torch.full_like(fig_t, col)
            t.index_put_((f, ans_t, qc_t, c), val, accumulate=True)
            t.index_put_((f, last, qc_t, c), val * nd, accumulate=True)
    if params.get("ddp") and dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)                      # :485-488
    breakdown_tensor += t       # the reference accumulates only under DDP (its evaluation never runs without it)
    return breakdown_tensor


def plotqa_evaluate(dataloader, dataset, params, eval_batch_size, dialog_encoder, group=None, with_histogram=True):
    """The evaluation loop (evaluation.py:199-386 without logging / CSV / plots).
    ``dataset`` needs ``get_ans_type(qa_id)`` (fig_dataloader.py:704-…) and may provide ``cut_batch_padding``.
    Returns (total_correct_tensor [6, 2], breakdown_tensor [5, 4, 3, 3], histogram [13]) on the device."""
    was_training = dialog_encoder.training
    dialog_encoder.eval()
    dev = torch.device(params["device"])
    breakdown = torch.zeros(5, 4, 3, 3, dtype=torch.float64, device=dev)
    total = torch.zeros(6, 2, dtype=torch.float64, device=dev)
    histogram = torch.zeros(13, dtype=torch.int64, device=dev)
    forced_mode = "_REGS" in str(params.get("qa_file", ""))
    with torch.no_grad():
        for batch in dataloader:
            (dataset.cut_batch_padding if hasattr(dataset, "cut_batch_padding") else cut_batch_padding)(batch)
            if batch["id"].shape[0] == 0:
                continue
            scores, out, err, terr = score_rows(dialog_encoder, batch, params, eval_batch_size)
            assert batch["tokens"].shape[0] == scores.shape[0] == out.shape[0] == err.shape[0] == terr.shape[0]   # :269
            forced = batch["gt_id"] if forced_mode else None
            answers, sel_out, sel_err, sel_terr, _ = select_answers(scores, out, err, terr, batch["num_ans"], forced)
            nsp_right, reg_right, reg_t_right, needs = correctness(answers, sel_err, sel_terr, batch)
            reduce_total_acc(total, needs, nsp_right, reg_right, reg_t_right, group)
            if "plotqa" in params["dataset"]:
                correct = nsp_right & (needs.logical_not() | reg_right)
                correct_t = nsp_right & (needs.logical_not() | reg_t_right)
                reduce_breakdown_table(dataset.get_ans_type, group, params, breakdown, batch, correct, correct_t, needs)
                if with_histogram:
                    reduce_histogram(histogram, sel_err[needs].view(-1), group)
    if was_training:
        dialog_encoder.train()
    return total, breakdown, histogram
