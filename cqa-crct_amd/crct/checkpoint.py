"""Checkpoint save / resume in the reference's own file format.

The reference writes, once per epoch and on rank 0 (CRCT/train.py:282-289)::

    torch.save({'model_state_dict': crct_model.module.state_dict(), 'scheduler_state_dict': scheduler.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(), 'iter_id': step_iter_id + 1},
               os.path.join(save_path, 'plotqa_encoder_%d_%d.ckpt' % (epoch, step)))

and reads it back in two modes (train.py:91-130, evaluation.py:22-66):

  * key intersection (default): the model takes every tensor whose key it knows (``bert_pretrained.*``);
    optimizer / scheduler start fresh;
  * ``params['continue']``: model + optimizer state (``state`` / ``param_groups``) + scheduler state + ``iter_id``;
    the epoch is parsed from the file name (third ``_`` field + 1).

A file written here loads in the reference and vice versa: identical keys, shapes, dtypes, one AdamW group per
tensor with ``step`` / ``exp_avg`` / ``exp_avg_sq`` per trained tensor (tests/golden/ckpt_schema.json, generated from
a real reference checkpoint, is what the tests compare against).  Tensors are written as independent CPU copies,
never as views of the flat HBM buffers.
"""
import os

import torch

from .optim import WarmupLinearScheduleNonZero


def checkpoint_file_name(epoch, step):
    return "plotqa_encoder_%d_%d.ckpt" % (epoch, step)                                   # train.py:282


def _unwrap(model):
    return getattr(model, "module", model)


def _cpu_copy(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().to("cpu", copy=True).contiguous()
    if isinstance(obj, dict):
        return {k: _cpu_copy(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu_copy(v) for v in obj)
    return obj


def checkpoint_dict(model, optimizer, scheduler, iter_id, loss_avg=None):
    """The dict of train.py:287-289 (plus the optional ``loss_avg`` the loader understands, train.py:124-125)."""
    if hasattr(optimizer, "synchronize"):
        optimizer.synchronize()                       # an overlapped update may still be running on its own stream
    out = {"model_state_dict": _cpu_copy(_unwrap(model).state_dict()),
           "scheduler_state_dict": _cpu_copy(scheduler.state_dict()),
           "optimizer_state_dict": _cpu_copy(optimizer.state_dict()),
           "iter_id": int(iter_id)}
    if loss_avg is not None:
        out["loss_avg"] = loss_avg
    return out


def save_checkpoint(save_path, model, optimizer, scheduler, epoch, step_iter_id, rank=0, loss_avg=None):
    """train.py:280-289: file ``plotqa_encoder_<epoch>_<step_iter_id + 1>.ckpt`` under ``save_path``, rank 0 only.
    Returns the path (also on the other ranks, which write nothing)."""
    path = os.path.join(save_path, checkpoint_file_name(epoch, step_iter_id + 1))
    if rank == 0:
        os.makedirs(save_path, exist_ok=True)
        torch.save(checkpoint_dict(model, optimizer, scheduler, step_iter_id + 1, loss_avg), path)
    return path


def load_model_weights(model, ckpt, strict_nonempty=True):
    """Key-intersection load (train.py:94-104, evaluation.py:31-41).  ``ckpt`` is a path or an already loaded dict.
    Returns the number of tensors transferred."""
    pretrained = _read(ckpt, _device_of(model))
    if "model_state_dict" in pretrained:
        pretrained = pretrained["model_state_dict"]
    target = _unwrap(model)
    model_dict = target.state_dict()
    picked = {k: v for k, v in pretrained.items() if k in model_dict}
    if strict_nonempty:
        assert len(picked.keys()) > 0                                                    # train.py:101
    model_dict.update(picked)
    target.load_state_dict(model_dict)
    return len(picked)


def resume(model, optimizer, ckpt_path, params, iters_per_epoch, scheduler_cls=WarmupLinearScheduleNonZero,
           restore_lr=False):
    """``-continue`` mode (train.py:106-130).  Returns ``(scheduler, start_iter_id, cont_epoch, loss_avg)``.

    Reference behaviour kept as is: the scheduler is rebuilt with ``last_epoch=iter_id`` -- its constructor takes
    one step, so the optimizer's learning rates become ``get_lr(iter_id + 1)`` -- and ``load_state_dict`` then
    restores the scheduler's own counters but not the optimizer's rates: the first resumed step runs one schedule
    tick ahead of an uninterrupted run.  ``restore_lr=True`` (not in the reference) writes the saved ``_last_lr``
    back into the param groups, which makes save -> resume exact."""
    pretrained = _read(ckpt_path, _device_of(model))
    cont_epoch = int(str(ckpt_path).split("/")[-1].split("_")[2]) + 1                   # train.py:107
    target = _unwrap(model)
    model_dict = target.state_dict()
    optimizer_dict = optimizer.state_dict()
    model_dict.update({k: v for k, v in pretrained["model_state_dict"].items() if k in model_dict})
    optimizer_dict.update({k: v for k, v in pretrained["optimizer_state_dict"].items() if k in optimizer_dict})
    target.load_state_dict(model_dict)
    optimizer.load_state_dict(optimizer_dict)
    scheduler = scheduler_cls(optimizer, warmup_steps=params["warmup"], min_lr=params["min_lr"],
                              t_total=iters_per_epoch * 20, last_epoch=pretrained["iter_id"])   # train.py:121-122
    scheduler.load_state_dict(pretrained["scheduler_state_dict"])
    if restore_lr:
        for g, lr in zip(optimizer.param_groups, pretrained["scheduler_state_dict"]["_last_lr"]):
            g["lr"] = lr
    return scheduler, pretrained["iter_id"], cont_epoch, pretrained.get("loss_avg")


def get_encoder(params, ckpt=None, config=None):
    """evaluation.py:22-66: build the encoder, load ``ckpt`` (either mode reads only the model weights there) and, in a
    multi-rank evaluation (``params['ddp'] and params['world_size'] > 1``, :56-61), wrap it as the reference does -- with
    ``FlatGradDDP`` in DistributedDataParallel's place (an initialised process group is the caller's, as there)."""
    from .model import VisualDialogEncoder
    enc = VisualDialogEncoder(params, config=config)
    if ckpt:
        load_model_weights(enc, ckpt)
    if params.get("ddp") and int(params.get("world_size", 1)) > 1:
        from .ddp import FlatGradDDP
        dev = _device_of(enc)
        enc = FlatGradDDP(enc, device_ids=[dev.index if getattr(dev, "index", None) is not None else 0], find_unused_parameters=True)
    return enc


def _device_of(model):
    core = _unwrap(model)
    core = getattr(core, "bert_pretrained", core)
    return core.flat_params.device if hasattr(core, "flat_params") else "cpu"


def _read(ckpt, device):
    if isinstance(ckpt, dict):
        return ckpt
    return torch.load(ckpt, map_location=device, weights_only=False)
