"""-m gpu: the reference's own training / evaluation call sequence, replayed line by line with the three imports swapped
(INTEGRATION.md section 2): CRCT/train.py:80-143 (construct from BERT-base -> get_optimizer -> scheduler -> DistributedDataParallel wrap),
:165-215 (autocast forward through the step adapter, the `.item()` statistics, GradScaler backward / step / update, scheduler),
:282-291 (`crct_model.module.state_dict()` checkpoint), :104-128 (`-continue` resume) and CRCT/evaluation.py:22-66 (`get_encoder`),
at the reference's own PlotQA shape (config/plotqa.json:5-6: 44 visual elements x 124 tokens, v_feature_size 1024)."""
import json
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from crct import config as C                                             # noqa: E402
from crct import synthetic as S                                          # noqa: E402
# ---- the swapped imports (train.py:13-16) ----
from crct.model import VisualDialogEncoder                               # noqa: E402
from crct.step_adapter import forward                                    # noqa: E402
from crct.optim import WarmupLinearScheduleNonZero, get_optimizer        # noqa: E402
from crct.ddp import FlatGradDDP                                         # noqa: E402      (for torch.nn.parallel.DistributedDataParallel)
from test_binding_cpu import bert_base_state_dict                        # noqa: E402


# Schedule of the test: warm-up 1 step, decay to the floor (min_lr 1.3e-5) by step 3.  The reference's resume sequence builds the scheduler
# with last_epoch = iter_id, which takes one scheduler step in the constructor (the optimizer's lr becomes lr(iter_id + 1)), and then
# restores last_epoch = iter_id from the checkpoint (train.py:118-121): a resumed run uses lr(iter_id + 1) twice and never lr(iter_id).
# Replayed verbatim that is what happens here too; on the floor both are min_lr, so the resumed run can be compared with the uninterrupted one.
T_TOTAL = 4


def _train_iteration(crct_model, batch, params, scaler, optimizer, scheduler, dist_grp, iter_id):
    """train.py:167-215, verbatim apart from names."""
    crct_model.train()
    num_regs = torch.sum(batch['needs_reg']).item()
    with torch.cuda.amp.autocast():
        loss, lm_loss, nsp_loss, img_loss, nsp_scores, regression, legend_loss = forward(crct_model, batch, params)
        reg_loss = regression[1][batch['needs_reg'].view(-1)]
        reg_5_right, reg_t_right = regression[3]
        reg_5_dist = regression[4][batch['needs_reg'].view(-1)]
        reg_loss = 0 if torch.isnan(reg_loss.mean()) else reg_loss.mean().item()
        reg_5_dist = 0 if torch.isnan(reg_5_dist.mean()) else reg_5_dist.mean().item()
        ddp_share_tensor = torch.tensor([loss.item(), lm_loss.mean().item(), nsp_loss.mean().item(), reg_loss, reg_5_dist, legend_loss.mean().item(),
                                         num_regs, reg_5_right, reg_t_right]).cuda()
        if params['ddp']:
            dist.all_reduce(ddp_share_tensor, op=torch.distributed.ReduceOp.SUM, group=dist_grp)
            ddp_share_tensor[:-3] = ddp_share_tensor[:-3] / params['world_size']
        if params['batch_multiply'] > 1:
            loss /= params['batch_multiply']
    scaler.scale(loss).backward()
    if iter_id % params['batch_multiply'] == 0:
        scaler.step(optimizer)
        optimizer.zero_grad()
        scaler.update()
        scheduler.step()
    return ddp_share_tensor.cpu()


def test_reference_loop_runs_unedited_after_the_import_swap(tmp_path):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    gpu = 0
    device = torch.device("cuda", gpu)
    torch.cuda.set_device(gpu)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)          # train.py:27-28
    try:
        # model_config: the shipped vilbert.json with the dropout probabilities zeroed, so that the resumed run can be compared with the
        # uninterrupted one (dropout masks are a function of a per-model call counter, which a resume restarts)
        cfg = json.loads(C.vilbert_config().to_json_string())
        for k in ("hidden_dropout_prob", "attention_probs_dropout_prob", "v_hidden_dropout_prob", "v_attention_probs_dropout_prob"):
            cfg[k] = 0.0
        cfg_path = str(tmp_path / "vilbert.json")
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
        bert = bert_base_state_dict(seed=3)
        bert_dir = tmp_path / "bert-base-uncased"
        bert_dir.mkdir()
        torch.save(bert, str(bert_dir / "pytorch_model.bin"))
        params = C.default_params(model_config=cfg_path, device=device, ddp=True, world_size=1, rank=0, batch_multiply=1, warmup=1, lr=2e-5, image_lr=2e-5,
                                  bert_pretrained=str(bert_dir), save_path=str(tmp_path), start_checkpoint="", **{"continue": False})
        assert (params["max_seq_len"], params["max_vis_features"]) == (124, 44)                 # config/plotqa.json:5-6
        batches = [S.make_batch(4, params["max_seq_len"], params["max_vis_features"], 1024, seed=50 + i, lengths=[124, 80, 61, 103], n_vis=[44, 31, 40, 44])
                   for i in range(5)]

        def build():
            crct_model = VisualDialogEncoder(params)                                             # train.py:80 (BERT-base from the local archive)
            crct_model.to(device)                                                                # :81
            crct_model.bert_pretrained.cls_dropout = 0.0                                         # (the head's hard-coded 0.1, vilbert.py:1045; see cfg above)
            optimizer = get_optimizer(params, crct_model)                                        # :84
            scheduler = WarmupLinearScheduleNonZero(optimizer, warmup_steps=params['warmup'], min_lr=params['min_lr'], t_total=T_TOTAL)      # :86
            return crct_model, optimizer, scheduler

        crct_model, optimizer, scheduler = build()
        sd = crct_model.state_dict()
        assert torch.equal(sd["bert_pretrained.bert.encoder.layer.3.output.LayerNorm.weight"].cpu(), bert["bert.encoder.layer.3.output.LayerNorm.gamma"])
        assert torch.equal(sd["bert_pretrained.bert.embeddings.word_embeddings.weight"].cpu(), bert["bert.embeddings.word_embeddings.weight"])
        assert sd["bert_pretrained.cls.predictions.decoder.weight"].data_ptr() == sd["bert_pretrained.bert.embeddings.word_embeddings.weight"].data_ptr()
        assert len(crct_model.pretrained_unexpected) == 5 and len(crct_model.pretrained_missing) == 561 - 202
        plain = crct_model
        crct_model = FlatGradDDP(crct_model, device_ids=[gpu], find_unused_parameters=True)     # :139-142
        dist_grp = dist.new_group(list(range(params['world_size'])))                             # :143
        crct_model.to(device)                                                                    # :145
        assert crct_model.module is plain
        optimizer.zero_grad()                                                                    # :148
        scaler = torch.cuda.amp.GradScaler()                                                     # :157
        stats = [_train_iteration(crct_model, batches[i], params, scaler, optimizer, scheduler, dist_grp, i) for i in range(3)]
        assert all(bool(torch.isfinite(s).all()) for s in stats) and 0.3 < float(stats[0][0]) < 3.0
        # ---- train.py:282-291
        file_name = 'plotqa_encoder_%d_%d.ckpt' % (0, 3)
        torch.save({'model_state_dict': crct_model.module.state_dict(), 'scheduler_state_dict': scheduler.state_dict(),
                    'optimizer_state_dict': optimizer.state_dict(), 'iter_id': 3}, os.path.join(params['save_path'], file_name))
        more = [_train_iteration(crct_model, batches[i], params, scaler, optimizer, scheduler, dist_grp, i) for i in (3, 4)]

        # ---- a second process's life: train.py:80-128 with params['continue']
        params2 = dict(params, start_checkpoint=os.path.join(params['save_path'], file_name), **{"continue": True})
        crct_model2, optimizer2, scheduler2 = build()
        pretrained_dict = torch.load(params2['start_checkpoint'], map_location=params2['device'])
        cont_epoch = int(params2['start_checkpoint'].split("/")[-1].split("_")[2]) + 1
        assert cont_epoch == 1
        model_dict = crct_model2.state_dict()
        optimizer_dict = optimizer2.state_dict()
        pretrained_dict_model = pretrained_dict['model_state_dict']
        pretrained_dict_optimizer = pretrained_dict['optimizer_state_dict']
        pretrained_dict_scheduler = pretrained_dict['scheduler_state_dict']
        pretrained_dict_model = {k: v for k, v in pretrained_dict_model.items() if k in model_dict}
        pretrained_dict_optimizer = {k: v for k, v in pretrained_dict_optimizer.items() if k in optimizer_dict}
        assert len(pretrained_dict_model) == 561
        model_dict.update(pretrained_dict_model)
        optimizer_dict.update(pretrained_dict_optimizer)
        crct_model2.load_state_dict(model_dict)
        optimizer2.load_state_dict(optimizer_dict)
        for state in optimizer2.state.values():
            for k, v in state.items():
                if isinstance(v, torch.Tensor):
                    state[k] = v.to(device)
        scheduler2 = WarmupLinearScheduleNonZero(optimizer2, warmup_steps=params2['warmup'], min_lr=params2['min_lr'], t_total=T_TOTAL,
                                                 last_epoch=pretrained_dict["iter_id"])
        scheduler2.load_state_dict(pretrained_dict_scheduler)
        assert scheduler2.last_epoch == 3 and optimizer2.param_groups[0]["lr"] == params2["min_lr"] == optimizer.param_groups[0]["lr"]
        crct_model2 = FlatGradDDP(crct_model2, device_ids=[gpu], find_unused_parameters=True)
        crct_model2.to(device)
        optimizer2.zero_grad()
        scaler2 = torch.cuda.amp.GradScaler()
        again = [_train_iteration(crct_model2, batches[i], params2, scaler2, optimizer2, scheduler2, dist_grp, i) for i in (3, 4)]
        for a, b in zip(more, again):          # same weights, same AdamW moments and step count, same schedule position: the same losses
            assert abs(float(a[0]) - float(b[0])) <= 1e-5 * abs(float(a[0])), (more, again)
            assert torch.equal(a[6:], b[6:])

        # ---- evaluation.py:22-66 (get_encoder), then a scoring forward (evaluation.py:258-262)
        dialog_encoder = VisualDialogEncoder(params2).to(device)
        pretrained_dict = torch.load(params2['start_checkpoint'], map_location=device)
        model_dict = dialog_encoder.state_dict()
        pretrained_dict_model = {k: v for k, v in pretrained_dict['model_state_dict'].items() if k in model_dict}
        model_dict.update(pretrained_dict_model)
        dialog_encoder.load_state_dict(model_dict)
        dialog_encoder = FlatGradDDP(dialog_encoder, device_ids=[params2['rank']], find_unused_parameters=True)
        dialog_encoder.to(device)
        dialog_encoder.eval()
        with torch.no_grad():
            loss, lm_loss, nsp_loss, img_loss, nsp_scores, regression = forward(dialog_encoder, batches[0], params2, output_nsp_scores=True, evaluation=True)
        assert loss is None and tuple(nsp_scores.shape) == (4, 2) and bool(torch.isfinite(nsp_scores).all()) and len(regression) == 5
    finally:
        dist.destroy_process_group()
