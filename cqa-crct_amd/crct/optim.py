"""Optimizer / LR-schedule surface of the reference (CRCT/utils.py:11-29, 228-249) on flat buffers.

``get_optimizer(params, model)`` returns a ``torch.optim.Optimizer`` with the reference's layout --
ONE param group per tensor in ``named_parameters()`` order, lr = ``params['lr']`` for BERT-base
language tensors else ``params['image_lr']``, weight decay 0 for names containing ``bias`` /
``LayerNorm.bias`` / ``LayerNorm.weight`` -- so ``optimizer.state_dict()`` / ``load_state_dict()``
round-trip with reference checkpoints (train.py:105-130, 284-291).  The update itself is ONE HIP
kernel over the flat parameter / gradient / moment buffers (crct_adamw_step), which also refreshes
the bf16 weight shadow; tensors that never receive a gradient are skipped, as torch.optim.AdamW
skips ``grad is None`` parameters.
"""
import contextlib
import json
import os

import torch
from torch.optim.lr_scheduler import _LRScheduler

from . import lib as L
from . import ops
from .events import order_streams
from .layout import NO_DECAY, is_language_weight


class WarmupLinearScheduleNonZero(_LRScheduler):
    """Linear warm-up to the base lr over ``warmup_steps``, then linear decay towards 0 at ``t_total``,
    floored at ``min_lr`` (utils.py:11-29)."""

    def __init__(self, optimizer, warmup_steps, t_total, min_lr=1.3e-5, last_epoch=-1):
        self.warmup_steps, self.t_total, self.min_lr = warmup_steps, t_total, min_lr
        super().__init__(optimizer, last_epoch=last_epoch)

    def get_lr(self):
        step = self.last_epoch
        if step < self.warmup_steps:
            f = float(step) / float(max(1, self.warmup_steps))
        else:
            f = max(0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))
        return [b * f if (b * f) > self.min_lr else self.min_lr for b in self.base_lrs]


def _crct_core(model):
    core = getattr(model, "bert_pretrained", model)
    core = getattr(core, "module", core)
    if not hasattr(core, "flat_params"):
        core = getattr(core, "bert_pretrained", core)
    return core


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics, one launch.  ``param_groups`` keep the caller's layout."""

    def __init__(self, param_groups, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        super().__init__(param_groups, defaults)
        self.core = _crct_core(model)
        core = self.core
        dev = core.flat_params.device
        self._m = torch.zeros_like(core.flat_params)
        self._v = torch.zeros_like(core.flat_params)
        self._step = 0
        byname = dict(core.named_parameters())
        self._used = [e for e in core.table if e.used]
        ids = {id(byname[e.name]): e for e in self._used}
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if id(p) in ids:
                    self._group_of[ids[id(p)].name] = gi
        self._segs = sorted([e for e in self._used if e.name in self._group_of], key=lambda e: e.offset)
        to_dev = lambda v, dt: torch.as_tensor(v, dtype=dt).to(dev)   # noqa: E731
        self._seg_off = to_dev([e.offset for e in self._segs], torch.int64)
        self._seg_len = to_dev([e.numel for e in self._segs], torch.int64)
        blk_seg, blk_off = ops.adamw_plan([e.numel for e in self._segs])
        self._blk_seg, self._blk_off = blk_seg.to(dev), blk_off.to(dev)
        self._lr_host = torch.empty(len(self._segs), dtype=torch.float32).pin_memory()
        self._wd_host = torch.empty(len(self._segs), dtype=torch.float32).pin_memory()
        self._lr_dev = torch.empty(len(self._segs), dtype=torch.float32, device=dev)
        self._wd_dev = torch.empty(len(self._segs), dtype=torch.float32, device=dev)
        self._last = None
        self._step_dev, self._amp_arg, self._amp_keep = None, None, None
        self._fp8_keep = None
        self._g16 = None                     # bf16 gradient source of the running update (data-parallel bf16 exchange), else None
        # crct/ddp.py: while THIS optimizer is alive and covers every gradient, a bf16 exchange need not write the weight gradients
        # back to fp32 (a weak reference: an optimizer that was built and discarded must not change what .grad holds)
        import weakref
        core._fused_optimizer = weakref.ref(self)
        # overlap mode: the update runs as one launch per engine backward-segment on its own stream, in first-use
        # order, each followed by an event; the next forward waits for segment s right before it needs it, so
        # AdamW (HBM-bound, ~1.2 ms) and the gradient memset overlap the next step's forward
        self.overlap = False
        self.overlap_workgroups = 256        # throttle of the overlapped launches (one workgroup per CU), 0 = full width
        self.full_width_first = 1            # this many of the first overlapped launches (embeddings first) run unthrottled
        self.launch_groups = 0               # > 0: the overlapped update in that many launches instead of one per backward segment
        self.fp8_transpose_workgroups = 256  # same for the transposed fp8 weight shadow that follows the update (fp8 backward)
        # the update zeroes every gradient element it has consumed; the zero_grad() that follows is then free.  Off by
        # default: torch optimizers leave .grad untouched in step() (a caller may still want to read it there)
        self.fuse_zero_grad = False
        # zero_grad() clears only what backward accumulates into; the Linear weight gradients are overwritten by the next
        # backward pass (no 0.96 GB fill, no read-modify-write in the weight-gradient GEMMs: -0.3 ms per step).  Their
        # .grad is stale (not zero) between zero_grad() and backward(); set False for the eager fill.
        self.lazy_zero_grad = True
        # early mode (with overlap): the update of a segment starts as soon as the backward pass has finished that
        # segment's gradients (the engine marks it with events), i.e. it overlaps the REST OF BACKWARD instead of the
        # next forward; identical arithmetic, only the start time on the GPU moves
        self.early = False
        self._upload_done = None
        self._opt_stream = None
        self._fp8_transposes = False
        self._seg_blocks = None
        self._events = None
        # expose the moments the way torch.optim.AdamW does (views of the flat buffers)
        for e in self._segs:
            p = byname[e.name]
            self.state[p] = dict(step=torch.tensor(0.0), exp_avg=self._m[e.offset:e.offset + e.numel].view(e.shape),
                                 exp_avg_sq=self._v[e.offset:e.offset + e.numel].view(e.shape))
        self._byname = byname

    def covers_every_gradient(self):
        """Does this optimizer update every tensor that receives a gradient?  (One built over a parameter subset leaves the rest to
        somebody who reads ``.grad``.)"""
        return len(self._segs) == len(self._used)

    def set_early(self, on=True):
        """Overlap the update with the rest of backward (needs ``overlap``); see ``early`` above."""
        self.early = bool(on)
        self.core.record_segment_events = bool(on)

    def _upload_hyper(self, stream=None):
        lrs = [self.param_groups[self._group_of[e.name]]["lr"] for e in self._segs]
        wds = [self.param_groups[self._group_of[e.name]]["weight_decay"] for e in self._segs]
        key = (tuple(lrs), tuple(wds))
        if key != self._last:
            if self._upload_done is not None:
                self._upload_done.synchronize()              # the pinned staging buffers are free again (long done in practice)
            self._lr_host.copy_(torch.tensor(lrs, dtype=torch.float32))
            self._wd_host.copy_(torch.tensor(wds, dtype=torch.float32))
            with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                self._lr_dev.copy_(self._lr_host, non_blocking=True)
                self._wd_dev.copy_(self._wd_host, non_blocking=True)
                if self._upload_done is None:
                    self._upload_done = torch.cuda.Event()
                self._upload_done.record()
            self._last = key

    def _launch(self, b0, b1, inv_scale, stream, max_workgroups=0):
        core, g0 = self.core, self.param_groups[0]
        L.check(L.load().crct_adamw_step(core.flat_params.data_ptr(), core.flat_grads.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                         core.flat_shadow.data_ptr(), self._seg_off.data_ptr(), self._seg_len.data_ptr(),
                                         self._lr_dev.data_ptr(), self._wd_dev.data_ptr(), self._blk_seg.data_ptr() + 4 * b0,
                                         self._blk_off.data_ptr() + 8 * b0, b1 - b0, g0["betas"][0], g0["betas"][1], g0["eps"],
                                         max(self._step, 1), L.ptr(inv_scale), self._amp_arg, self._fp8_arg(), int(max_workgroups),
                                         int(self.fuse_zero_grad), L.ptr(self._g16), stream), "adamw_step")

    # ---- torch.amp.GradScaler (train.py:157,208-212).  ``scaler.step(optimizer)`` sees ``_step_supports_amp_scaling`` and
    # hands over ``optimizer.grad_scale`` / ``optimizer.found_inf`` (device scalars) instead of unscaling 524 gradient views
    # itself; the update kernel divides by the scale, skips the step on inf / nan and takes its bias corrections from a
    # device-side step counter that only advances on real steps -- no host sync anywhere.
    _step_supports_amp_scaling = True

    def _amp_begin(self):
        import ctypes as C
        scale, found = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        if scale is None and found is None:
            if self._step_dev is not None:                      # scaler went away: back to the host counter
                self._step = int(self._step_dev.item())
                self._step_dev = None
            self._amp_arg = None
            return False
        dev = self.core.flat_params.device
        if self._step_dev is None:
            self._step_dev = torch.tensor([self._step], dtype=torch.int32, device=dev)
        found32 = found.to(device=dev, dtype=torch.float32).reshape(1) if found is not None else None
        scale32 = scale.to(device=dev, dtype=torch.float32).reshape(1) if scale is not None else None
        L.check(L.load().crct_adamw_advance(self._step_dev.data_ptr(), L.ptr(found32), L.current_stream()), "adamw_advance")
        st = L.AmpState()
        st.grad_scale, st.found_inf, st.step = L.ptr(scale32), L.ptr(found32), self._step_dev.data_ptr()
        self._amp_keep = (scale32, found32, st)
        self._amp_arg = C.byref(st)
        return True

    def _fp8_arg(self):
        """CrctFp8Shadow of the model's e4m3 weight shadow (None unless the model runs the fp8 forward)."""
        core = self.core
        st = getattr(core, "_fp8", None)
        if not getattr(core, "fp8", False) or st is None or not st["weights"]:
            return None
        if self._fp8_keep is None or self._fp8_keep[0] is not st:
            import bisect
            import ctypes as C
            by_start = sorted(range(len(st["weights"])), key=lambda i: st["weights"][i][0])      # the engine lists them in model order, not by offset
            starts = [st["weights"][i][0] for i in by_start]
            slots, t_in, t_base, t_ld = [], [], [], []
            w_in = st["w_in"].tolist()
            for e in self._segs:                         # AdamW segments are parameter tensors: a fused QKV weight spans three
                k = bisect.bisect_right(starts, e.offset) - 1
                k = by_start[k] if k >= 0 else -1
                inside = k >= 0 and e.offset + e.numel <= st["weights"][k][0] + st["weights"][k][1]
                slots.append(k if inside else -1)
                # transposed shadow (fp8 backward): kept by the update itself, in whole 64 x 64 tiles, where the segment is a band
                # of whole rows of the weight (the weight itself, or the Q / K / V part of a fused QKV weight)
                ok = inside and st.get("qt") is not None and st["transposed"][k]
                if ok:
                    w_off, w_num = st["weights"][k]
                    n_in = w_in[k]
                    ok = n_in % 64 == 0 and (e.offset - w_off) % n_in == 0 and e.numel % (64 * n_in) == 0 and ((e.offset - w_off) // n_in) % 16 == 0
                t_in.append(w_in[k] if ok else 0)
                t_base.append(st["weights"][k][0] + (e.offset - st["weights"][k][0]) // w_in[k] if ok else 0)
                t_ld.append(st["weights"][k][1] // w_in[k] if ok else 0)
            dev = core.flat_params.device
            seg_slot = torch.tensor(slots, dtype=torch.int32, device=dev)
            seg_in = torch.tensor(t_in, dtype=torch.int32, device=dev)
            seg_t_base = torch.tensor(t_base, dtype=torch.int64, device=dev)
            seg_t_ld = torch.tensor(t_ld, dtype=torch.int32, device=dev)
            sh = L.Fp8Shadow()
            sh.q, sh.seg_slot, sh.scale, sh.amax = st["q"].data_ptr(), seg_slot.data_ptr(), st["w_scale"].data_ptr(), st["w_amax"].data_ptr()
            # every weight that has a transposed copy must be covered, or the separate launch stays (crct_fp8_transpose_weights)
            if set(s for s in slots if s >= 0) != set(range(len(st["weights"]))):
                raise RuntimeError("fused AdamW: %d of the model's %d fp8-shadowed weights are not covered by optimizer segments; their "
                                   "shadow would go stale" % (len(st["weights"]) - len(set(s for s in slots if s >= 0)), len(st["weights"])))
            covered = [0] * len(st["weights"])
            for s, t, e in zip(slots, t_in, self._segs):
                if s >= 0 and t > 0:
                    covered[s] += e.numel
            fused = st.get("qt") is not None and all(covered[k] == st["weights"][k][1] for k in range(len(covered)) if st["transposed"][k])
            if fused:
                sh.qt, sh.seg_in, sh.seg_t_base, sh.seg_t_ld = st["qt"].data_ptr(), seg_in.data_ptr(), seg_t_base.data_ptr(), seg_t_ld.data_ptr()
            self._fp8_transposes = fused
            self._fp8_keep = (st, seg_slot, sh, C.byref(sh), seg_in, seg_t_base, seg_t_ld)
        return self._fp8_keep[3]

    def _fp8_before_update(self, stream):
        """Delayed scaling of the weight shadow: the scales this update quantises with come from the amax the previous update saw."""
        if self._fp8_arg() is not None:
            st = self.core._fp8
            # the window is counted on the host per CALL (under GradScaler the host step counter is frozen: the device counter
            # takes over), and a step the scaler skips leaves shadow AND scales alone
            st["w_updates"] = st.get("w_updates", 0) + 1
            found = self._amp_keep[1] if (self._amp_arg is not None and self._amp_keep is not None) else None
            L.check(L.load().crct_fp8_update_scales(st["w_scale"].data_ptr(), st["w_amax"].data_ptr(), len(st["weights"]),
                                                    int(st["w_updates"] % self.core.FP8_AMAX_WINDOW == 0), L.ptr(found), 448.0, stream),
                    "fp8_update_scales")

    def _launch_groups(self, order):
        """``order`` (segments in first-use order) cut into ``launch_groups`` runs of similar block counts, the first segment alone."""
        key = (tuple(order), self.launch_groups)
        if getattr(self, "_groups_key", None) != key:
            sizes = [self._seg_blocks[i][1] - self._seg_blocks[i][0] for i in order]
            groups, rest = [[order[0]]], order[1:]
            n_rest = max(self.launch_groups - 1, 1)
            target = (sum(sizes[1:]) + n_rest - 1) // n_rest
            cur, acc = [], 0
            for i, sz in zip(rest, sizes[1:]):
                cur.append(i)
                acc += sz
                if acc >= target and len(groups) < self.launch_groups - 1:
                    groups.append(cur)
                    cur, acc = [], 0
            if cur:
                groups.append(cur)
            for g in groups:                     # block ranges of a run must be adjacent (the flat buffer is in reverse first-use order)
                spans = sorted(self._seg_blocks[i] for i in g if self._seg_blocks[i][1] > self._seg_blocks[i][0])
                if any(a[1] != b[0] for a, b in zip(spans, spans[1:])):
                    raise RuntimeError("optimizer launch groups: segments of a group are not adjacent in the block table")
            self._groups_key, self._groups = key, groups
            self._group_events = [None] * len(self._seg_blocks)
        return self._groups

    def _plan_overlap(self):
        """Block ranges of the optimizer's table per engine backward-segment (both are sorted by flat offset)."""
        eng = self.core._engine
        if eng is None:
            return False
        import bisect
        blk_seg = self._blk_seg.cpu().tolist()
        first_blk = {}
        for i, sgi in enumerate(blk_seg):
            first_blk.setdefault(sgi, i)
        offs = [e.offset for e in self._segs]
        self._seg_blocks = []
        for lo, hi in eng.segments:
            s0, s1 = bisect.bisect_left(offs, lo), bisect.bisect_left(offs, hi)
            b0 = first_blk[s0] if s0 < len(offs) and s0 in first_blk else len(blk_seg)
            b1 = first_blk[s1] if s1 < len(offs) and s1 in first_blk else len(blk_seg)
            self._seg_blocks.append((b0, b1))
        covered = sum(b1 - b0 for b0, b1 in self._seg_blocks)
        if covered != len(blk_seg):
            raise RuntimeError("optimizer overlap plan does not cover every block (%d of %d)" % (covered, len(blk_seg)))
        self._opt_stream = self.core.aux_stream()        # shared with the data-parallel exchange (crct/ddp.py)
        from .events import DeviceEvent
        self._events = [DeviceEvent() for _ in eng.segments]
        return True

    def _follow_device(self):
        """The model was moved (``model.to('cuda:1')`` after the optimizer was built): the moments and tables follow, and
        everything bound to the old device -- update stream, segment events, fp8 shadow descriptor -- is made again."""
        dev = self.core.flat_params.device
        if self._m.device == dev:
            return
        old_m, old_v = self._m, self._v
        self._m, self._v = old_m.to(dev), old_v.to(dev)
        for e in self._segs:
            st = self.state[self._byname[e.name]]
            st["exp_avg"] = self._m[e.offset:e.offset + e.numel].view(e.shape)
            st["exp_avg_sq"] = self._v[e.offset:e.offset + e.numel].view(e.shape)
        for k in ("_seg_off", "_seg_len", "_blk_seg", "_blk_off", "_lr_dev", "_wd_dev"):
            setattr(self, k, getattr(self, k).to(dev))
        self._last = None
        self._upload_done = None
        self._fp8_keep = self._events = self._opt_stream = self._seg_blocks = None
        self._step_dev = self._step_dev.to(dev) if self._step_dev is not None else None

    @torch.no_grad()
    def step(self, closure=None, inv_scale=None):
        loss = closure() if closure is not None else None
        core = self.core
        self._follow_device()
        amp = self._amp_begin()
        if not amp:
            self._step += 1
        # data-parallel bf16 exchange: the all-reduced gradients live in the communication buffer (crct/ddp.py), not in .grad
        ddp = getattr(core, "_ddp", None)
        self._g16 = ddp.grad_source() if ddp is not None else None
        if self._g16 is not None and (amp or inv_scale is not None):
            raise RuntimeError("GradScaler / unscaling reads the fp32 .grad views: construct FlatGradDDP(..., materialize_grads=True) "
                               "(or grad_dtype=torch.float32) when training with a GradScaler")
        if self.overlap and not amp and (self._seg_blocks is not None or self._plan_overlap()):
            cur = torch.cuda.current_stream()
            self._opt_stream = core.aux_stream()          # the engine's auxiliary stream (re-fetched: an engine rebuilt for a larger batch has new streams)
            done = None
            if self.early and inv_scale is None:
                done = core.take_segment_done_events()
                if done is None and ddp is not None and ddp.last_exchange is not None:
                    done = ddp.segment_waits()       # AdamW of a bucket behind THAT bucket's all-reduce, not behind all of them
            n = len(self._seg_blocks)
            if done is None:
                self._upload_hyper()
                # gradients (and the hyper-parameter upload) are final; after an all-reduce across GPUs they were written by
                # peer devices: system-scope ordering then
                ddp = getattr(core, "_ddp", None)
                order_streams(cur, self._opt_stream, system=ddp is not None and getattr(ddp, "world", 1) > 1)
                order = range(n - 1, -1, -1)                  # first-use order: embeddings ... heads
            else:
                self._upload_hyper(self._opt_stream)          # not behind the backward pass that `cur` still runs
                order = range(n)                              # the order backward finishes them: heads ... embeddings
            self._fp8_before_update(self._opt_stream.cuda_stream)
            if done is None and self.launch_groups > 0:
                # fewer, larger launches: consecutive segments (adjacent block ranges) share one launch and one event; the first
                # group is the first segment alone, so that the next forward can start at once
                for members in self._launch_groups(list(order)):
                    b0 = min(self._seg_blocks[i][0] for i in members)
                    b1 = max(self._seg_blocks[i][1] for i in members)
                    if b1 > b0:
                        self._launch(b0, b1, inv_scale, self._opt_stream.cuda_stream, self.overlap_workgroups)
                    ev = self._events[members[0]]
                    ev.record(self._opt_stream)
                    for i in members[1:]:
                        self._group_events[i] = ev
                    self._group_events[members[0]] = ev
                core._param_events = list(self._group_events)
                core._opt_stream = self._opt_stream
                order = ()
            launched = 0
            for sgi in order:
                b0, b1 = self._seg_blocks[sgi]
                if done is not None:
                    for w in done[sgi]:
                        w(self._opt_stream)
                if b1 > b0:
                    # the first launches in first-use order (embeddings, first layers) run while the next forward is still waiting
                    # for them: nothing to share the chip with yet, so no throttle (full_width_first = how many)
                    wide = done is None and launched < int(self.full_width_first)
                    self._launch(b0, b1, inv_scale, self._opt_stream.cuda_stream, 0 if wide else self.overlap_workgroups)
                    launched += 1
                self._events[sgi].record(self._opt_stream)
            if done is not None:
                order_streams(cur, self._opt_stream)          # the gradient memset that follows must not pass backward's tail
            if not (done is None and self.launch_groups > 0):
                core._param_events = self._events            # the next forward waits segment by segment
            core._opt_stream = self._opt_stream               # ... and the next backward for the whole stream
        else:
            self._upload_hyper()
            self._fp8_before_update(L.current_stream())
            self._launch(0, self._blk_seg.numel(), inv_scale, L.current_stream())
        if getattr(core, "fp8_backward", False) and self._fp8_arg() is not None:
            # the update rewrote the e4m3 weight shadow: its transposed copy (fp8 data gradients) follows on the same stream
            beside = self.overlap and not amp and self._opt_stream is not None
            side = self._opt_stream.cuda_stream if beside else L.current_stream()
            if not self._fp8_transposes:                # normally the update kernel has written the transposed copy itself
                core._fp8_transpose(side, self.fp8_transpose_workgroups if beside else 0)
            core._fp8_update_grad_scales(side)          # the scales the NEXT backward pass quantises its gradients with
        core.note_params_updated_natively()
        # under GradScaler the kernel returns at once on a skipped step and zeroes nothing: only without it the clear is certain
        self._grads_cleared = bool(self.fuse_zero_grad) and not amp
        if self._grads_cleared:
            core._grads_dirty = False                         # until the next backward pass
        return loss

    def synchronize(self):
        """Order the current stream after an in-flight overlapped update (before reading weights outside forward)."""
        if self._opt_stream is not None:
            order_streams(self._opt_stream, torch.cuda.current_stream())

    def zero_grad(self, set_to_none=True):
        # one memset; .grad views stay attached (set_to_none would only force a re-attach next step)
        if getattr(self, "_grads_cleared", False) and not self.core._grads_dirty:
            return                                            # step() has already zeroed them (fuse_zero_grad)
        if self.overlap and self._opt_stream is not None:
            with torch.cuda.stream(self._opt_stream):        # after the update that is still reading the gradients
                self.core.zero_flat_grads(lazy=self.lazy_zero_grad)
        else:
            self.core.zero_flat_grads(lazy=self.lazy_zero_grad)

    def state_dict(self):
        if self._step_dev is not None:
            self._step = int(self._step_dev.item())
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        idx = 0
        pos_to_param = {}
        for g, sg in zip(self.param_groups, groups):
            for p, sid in zip(g["params"], sg["params"]):
                pos_to_param[sid] = p
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
        step = 0
        for sid, st in state_dict["state"].items():
            p = pos_to_param[sid]
            mine = self.state.get(p)
            if mine is None:
                continue
            mine["exp_avg"].copy_(st["exp_avg"])
            mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self._step = step
        self._step_dev = None
        self._last = None


def get_optimizer(params, dialog_encoder, language_weights_json=None):
    """utils.py:228-249 with the fused kernel underneath.  ``language_weights_json`` may point at the
    reference's ``config/language_weights.json``; by default its membership rule is applied
    (crct.layout.is_language_weight)."""
    listed = None
    path = language_weights_json or os.path.join("config", "language_weights.json")
    if language_weights_json or os.path.exists(path):
        with open(path) as f:
            listed = set(json.load(f))
    groups = []
    for key, value in dict(dialog_encoder.named_parameters()).items():
        if not value.requires_grad:
            continue
        bare = key[len("bert_pretrained."):] if key.startswith("bert_pretrained.") else key
        lang = (key in listed) if listed is not None else is_language_weight(bare)
        lr = params["lr"] if lang else params["image_lr"]
        wd = 0 if any(nd in key for nd in NO_DECAY) else params["wd"]
        groups.append({"params": [value], "lr": lr, "weight_decay": wd})
    return FusedAdamW(groups, dialog_encoder, lr=params["lr"])
