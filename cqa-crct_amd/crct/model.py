"""Host-side mirror of the reference model surface for the CRCT hot path.

  ``VisualDialogEncoder(params)``           <- CRCT/backbone/encoder_decorator.py:9-54
  ``CrctModel`` (``.bert_pretrained``)      <- BertForMultiModalPreTraining, CRCT/backbone/vilbert.py:1499-1661

Same constructor arguments, same ``forward`` keyword arguments / return tuples, same
``state_dict`` keys, shapes and order (561 entries, incl. the tied ``cls.predictions.decoder.weight``),
same ``named_parameters()`` order (560), so the reference's optimizer construction, checkpoints
(``train.py:284-291``) and step adapter keep working.  Underneath there is no torch compute: all
parameters are views into ONE flat fp32 HBM buffer (+ a bf16 shadow, + a flat fp32 gradient buffer)
and ``forward`` / ``backward`` are single calls into the native step engine (hand-written gfx950
kernels, ``include/crct_hip.h``).  The module refuses to run without the HIP library or off-GPU.
"""
import ctypes as C
import math
import os

import torch
from torch import nn

from . import lib as L
from . import ops
from .config import BertConfig
from .engine import StepEngine
from .layout import parameter_table


class _Node(nn.Module):
    """Name-only container used to reproduce the reference's dotted parameter names."""

    def forward(self, *a, **k):   # pragma: no cover
        raise RuntimeError("structural node of the CRCT parameter tree; call the model instead")


class _StepFn(torch.autograd.Function):
    """Autograd bridge.  Differentiable outputs: the COMBINED loss (0-dim, = nsp_coeff * nsp + reg_coeff * mean_B reg_loss,
    computed by the head kernel, encoder_decorator.py:144-153), nsp_loss [1] and the per-row regression loss [B]; backward
    hands their upstream gradients (device tensors, no sync) to the engine, which accumulates every parameter gradient
    into the flat buffer (``param.grad`` are views of it).  The usual case -- only ``loss`` is differentiated -- costs no
    torch kernel besides autograd's own seed: the head kernel scales its built-in seeds by the upstream scalar."""

    @staticmethod
    def forward(ctx, anchor, model, tensors, step):
        # (fp8 data gradients only: the forward pass is the plain bf16 one -- no fp8 state, no copies -- the backward pass gets it)
        fwd_step = {k: v for k, v in step.items() if k != "fp8"} if step.get("fp8_backward_only") else step
        model._engine.forward(model._flat_p, model._flat_b16, tensors, fwd_step)
        ctx.model, ctx.tensors, ctx.step = model, tensors, step
        ctx.set_materialize_grads(False)
        logits, reg_all, stats = model._engine.snapshot(tensors["tokens"].shape[0])    # one copy; the engine reuses its buffer
        ctx.mark_non_differentiable(logits, reg_all, stats)
        # the differentiable scalar is a tensor of its own (4 bytes): the reference loop modifies it in place
        # (train.py:204-205 ``loss /= params['batch_multiply']``), which autograd forbids on a view output of a multi-output node
        return stats[0].clone(), stats[1:2], reg_all[1], logits, reg_all, stats     # the rest: views of the snapshot, no copies

    @staticmethod
    def backward(ctx, g_loss, g_nsp, g_reg, _gl, _gr, _gs):
        model = ctx.model
        step = dict(ctx.step)
        if g_nsp is None and g_reg is None:
            if g_loss is None:
                return None, None, None, None
            step["g_loss"] = g_loss.reshape(1).float()                 # views / no-ops for the fp32 scalar autograd hands over
        else:                                                          # a caller combining nsp_loss / reg_loss itself
            B = ctx.tensors["tokens"].shape[0]
            dev = model._flat_p.device
            gl = g_loss.reshape(1).float() if g_loss is not None else torch.zeros(1, device=dev)
            gn = g_nsp.reshape(1).float() if g_nsp is not None else torch.zeros(1, device=dev)
            gr = g_reg.reshape(B).float() if g_reg is not None else torch.zeros(B, device=dev)
            step["g_nsp"] = (gn + gl * step["nsp_coeff"]).contiguous()
            step["g_reg"] = (gr + gl * (step["reg_coeff"] / B)).contiguous()
        model._run_backward(ctx.tensors, step)
        return None, None, None, None


class SequenceMask(object):
    """What the step adapter knows about the text key mask -- ``arange(T) < sep_indices[hist_len] + 1``
    (encoder_decorator.py:118-120) -- handed to the model as ``attention_mask`` instead of the materialised tensor: the
    engine builds the uint8 key mask itself (one launch).  ``materialize()`` gives the reference's tensor."""

    def __init__(self, sep_indices, hist_len, max_len):
        self.sep_indices, self.hist_len, self.max_len = sep_indices, hist_len, int(max_len)

    def materialize(self):
        lengths = torch.gather(self.sep_indices, 1, self.hist_len.view(-1, 1)).squeeze(1) + 1
        rng = torch.arange(0, self.max_len, device=lengths.device).long()
        return rng.unsqueeze(0) < lengths.unsqueeze(1)


class CrctModel(nn.Module):
    def __init__(self, config, params=None):
        super().__init__()
        if not isinstance(config, BertConfig):
            raise ValueError("Parameter config in `CrctModel(config)` should be an instance of class `BertConfig`.")
        if params.get("CE_REG") or params.get("binary_answers") or params.get("dataset", "plotqa") not in ("plotqa", "plotqa_colorless"):
            raise NotImplementedError("only the PlotQA regression path (PlotQA_Regressor_v20) is built (SURVEY.md section 2)")
        self.config, self.params = config, params
        device = torch.device(params.get("device", "cuda"))
        if device.type != "cuda":
            raise RuntimeError("CrctModel needs an MI355X (params['device']=%s): the CRCT step has no CPU path" % device)
        self.table, self.total = parameter_table(config, params)
        self._flat_p = torch.zeros(self.total, device=device)
        self._flat_g = torch.zeros(self.total, device=device)
        self._flat_b16 = torch.zeros(self.total, device=device, dtype=torch.bfloat16)
        self._shadow_ver = -1
        self._rebound = None
        self._const_zeros = None
        # fp8 forward (BASELINE configs[4], params['fp8']): e4m3 shadow of the QKV / FFN weights + per-tensor delayed scaling
        self.fp8 = bool(params.get("fp8", False)) if params else False
        # ... and the data-gradient GEMMs of the FFN / attention-output Linears from e5m2 gradients and a transposed e4m3 weight
        # shadow (params['fp8_backward'] = False keeps the round-2 behaviour: fp8 forward, bf16 backward)
        self.fp8_backward = self.fp8 and bool(params.get("fp8_backward", True))
        # ... and the FFN weight gradients from the same fp8 copies (params['fp8_wgrad'] = False: bf16 weight gradients)
        self.fp8_wgrad = self.fp8_backward and bool(params.get("fp8_wgrad", True))
        # params['fp8_forward'] = False (round 4): the forward GEMMs stay on the bf16 operands -- the producers still write the e4m3 /
        # scale state the fp8 BACKWARD GEMMs read (the engine's calibration form of the forward, CrctStepCfg.fp8 = 2, kept on).  What
        # costs an fp8 step its gradient fidelity is the e4m3 rounding of the FORWARD (oracle emulation, tools/lab/mx_emulation.py:
        # median gradient cosine 0.90 forward-only, 0.88 with everything; block scaling does not change it), while fp8 data and weight
        # gradients behind an unrounded forward cost ~0.01: this mode keeps most of the speed (backward is 2/3 of the GEMM work).
        self.fp8_forward = bool(params.get("fp8_forward", True)) if params else True
        # params['residual_fp32'] (default on, round 6): the encoder's residual stream is carried in fp32 like the reference's autocast path
        # carries it -- pre-LayerNorm sums and the residual copy of every LayerNorm output; GEMM operands stay bf16 (CrctStepCfg.residual_fp32).
        # False = rounds 1 - 5: both stored as bf16, which is what cost the bf16 path most of its gradient-cosine deficit.
        self.residual_fp32 = bool(params.get("residual_fp32", True)) if params else True
        self._fp8 = None
        self._entries = {e.name: e for e in self.table}
        self._build_tree()
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._engine = None
        self._seed, self._calls = int(params.get("seed", 0)) * 1000003 + 12345, 0
        self.cls_dropout = 0.1                       # vilbert.py:1045
        self.sync_stats = True                       # reg[3] as python ints (host sync) like the reference
        self._ddp = None
        self._param_events = None                    # set by FusedAdamW in overlap mode
        self._seg_done = None
        self._grads_dirty = True
        self._wgrad_overwrite_next = False       # set by a lazy clear: the next backward pass writes the owned weight gradients
        self._backward_passes = 0
        self._lazy_plan_key, self._lazy_plan = None, None
        self._grad_waits = None
        self.record_segment_events = False           # set by FusedAdamW's early mode
        self._opt_stream = None
        self.init_weights(int(params.get("seed", 0)))
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate_shadow())

    # ------------------------------------------------------------------ parameter tree
    def _build_tree(self):
        word = None
        for e in self.table:
            view = self._flat_p[e.offset:e.offset + e.numel].view(e.shape)
            p = nn.Parameter(view, requires_grad=True)
            node = self
            parts = e.name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], p)
            if e.used:
                p.grad = self._flat_g[e.offset:e.offset + e.numel].view(e.shape)
            if e.name == "bert.embeddings.word_embeddings.weight":
                word = p
        # tied LM decoder weight (vilbert.py:1029): second state_dict key, same Parameter
        node = self._modules["cls"]._modules["predictions"]
        node.add_module("decoder", _Node())
        node._modules["decoder"].register_parameter("weight", word)

    def _params_by_name(self):
        return dict(self.named_parameters())

    @torch.no_grad()
    def init_weights(self, seed=0):
        """init_bert_weights (vilbert.py:1099-1110): N(0, initializer_range) for Linear / Embedding weights,
        zero biases, LayerNorm = (1, 0); the regressor keeps nn.Linear's default init (it is created
        after ``self.apply(self.init_bert_weights)``, vilbert.py:1510-1523)."""
        g = torch.Generator(device=self._flat_p.device).manual_seed(seed)
        for e in self.table:
            v = self._flat_p[e.offset:e.offset + e.numel]
            if e.name.startswith("regressor."):
                fan_in = e.shape[1] if len(e.shape) == 2 else self._entries[e.name[:-4] + "weight"].shape[1]
                bound = 1.0 / math.sqrt(fan_in)
                v.uniform_(-bound, bound, generator=g)
            elif "LayerNorm" in e.name:
                v.fill_(1.0 if e.name.endswith("weight") else 0.0)
            elif e.name.endswith("bias"):
                v.zero_()
            else:
                v.normal_(0.0, self.config.initializer_range, generator=g)
        self._invalidate_shadow()

    def _invalidate_shadow(self):
        self._shadow_ver = -1

    def _param_version(self):
        v = self._flat_p._version
        if self._rebound is not None:                # see _rebind: in-place updates through re-pointed Parameters
            v = (v, sum(p._version for p in self._rebound))
        return v

    def _refresh_shadow(self):
        v = self._param_version()
        if self._shadow_ver != v:
            ops.cast_bf16(self._flat_p, out=self._flat_b16)
            self._shadow_ver = v
            if self._fp8 is not None:
                self._fp8_requantize()

    # ------------------------------------------------------------------ fp8 forward state
    def _fp8_state(self, eng):
        """Device state of the fp8 forward, created with the first engine: e4m3 weight shadow (same element offsets as the
        flat fp32 buffer), weight scales / amax per shadowed weight, activation scales / amax per producer site."""
        if self._fp8 is None:
            n_sites, weights = eng.fp8_layout()
            dev = self._flat_p.device
            st = dict(n_sites=n_sites, weights=weights,
                      q=torch.zeros(self.total, dtype=torch.uint8, device=dev),
                      w_scale=torch.ones(max(len(weights), 1), device=dev), w_amax=torch.zeros(max(len(weights), 1) * L.FP8_AMAX_LANES, device=dev),
                      a_scale=torch.ones(max(n_sites, 1), device=dev), a_amax=torch.zeros(max(n_sites, 1) * L.FP8_AMAX_LANES, device=dev),
                      calibrated=False)
            # chunk table over the shadowed weights themselves (one segment per weight, slot = its index)
            lens = [n for _, n in weights]
            blk_seg, blk_off = ops.adamw_plan(lens) if lens else (torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int64))
            st["seg_off"] = torch.tensor([o for o, _ in weights], dtype=torch.int64, device=dev)
            st["seg_len"] = torch.tensor(lens, dtype=torch.int64, device=dev)
            st["seg_slot"] = torch.arange(len(weights), dtype=torch.int32, device=dev)
            st["blk_seg"], st["blk_off"] = blk_seg.to(dev), blk_off.to(dev)
            # fp8 backward: transposed shadow (same byte offsets, every weight stored [in][out]), gradient scale sites
            by_off = {e.offset: e for e in self.table}
            outs, ins, begins, nt, transposed = [], [], [], 0, []
            for off, num in weights:
                n_in = by_off[off].shape[1]
                outs.append(num // n_in)
                ins.append(n_in)
                begins.append(nt)
                # every shadowed weight has a transposed copy (the fp8 data gradients; a fused QKV weight [3H][H] is one of them)
                transposed.append(True)
                if transposed[-1]:
                    nt += ((num // n_in + 63) // 64) * ((n_in + 63) // 64)
            st["transposed"] = transposed
            st["qt"] = torch.zeros(self.total, dtype=torch.uint8, device=dev) if self.fp8_backward else None
            st["w_out"] = torch.tensor(outs, dtype=torch.int32, device=dev)
            st["w_in"] = torch.tensor(ins, dtype=torch.int32, device=dev)
            st["tile_begin"] = torch.tensor(begins, dtype=torch.int64, device=dev)
            st["n_tiles"] = nt
            n_g = eng.lib.crct_engine_fp8_grad_sites(eng.handle)
            st["n_gsites"] = n_g
            st["g_scale"] = torch.ones(max(n_g, 1), device=dev)
            st["g_amax"] = torch.zeros(max(n_g, 1) * L.FP8_AMAX_LANES, device=dev)
            st["bwd_calibrated"] = False
            self._fp8 = st
            self._fp8_requantize()
        return self._fp8

    def _fp8_requantize(self):
        """Exact per-tensor quantisation of the current fp32 weights (start-up, load_state_dict, foreign optimizers); the
        fused AdamW keeps the shadow current by itself afterwards."""
        st = self._fp8
        if not st["weights"]:
            return
        L.check(L.load().crct_fp8_quantize_weights(self._flat_p.data_ptr(), st["q"].data_ptr(), st["seg_off"].data_ptr(),
                                                   st["seg_len"].data_ptr(), st["seg_slot"].data_ptr(), st["blk_seg"].data_ptr(),
                                                   st["blk_off"].data_ptr(), st["blk_seg"].numel(), st["w_scale"].data_ptr(),
                                                   st["w_amax"].data_ptr(), len(st["weights"]), L.current_stream()), "fp8_quantize_weights")
        self._fp8_transpose(L.current_stream())

    def _fp8_update_grad_scales(self, stream):
        """Gradient scales of the fp8 backward from the maxima the last backward pass collected (delayed scaling).  The fused
        AdamW calls it on its own stream, off the critical path; without it the next backward pass does it first thing."""
        st = self._fp8
        if st is None or not self.fp8_backward or not st.get("bwd_calibrated") or st["n_gsites"] <= 0:
            return
        st["g_updates"] = st.get("g_updates", 0) + 1
        L.check(L.load().crct_fp8_update_scales(st["g_scale"].data_ptr(), st["g_amax"].data_ptr(), st["n_gsites"],
                                                int(st["g_updates"] % self.FP8_AMAX_WINDOW == 0), None, 57344.0, stream),
                "fp8_update_scales (gradients)")
        st["g_scales_fresh"] = True

    def _fp8_transpose(self, stream, max_workgroups=0):
        """The transposed e4m3 weight shadow the fp8 data-gradient GEMMs read, rebuilt from the shadow (one launch, 2 bytes per
        shadowed weight element); called whenever the shadow has been rewritten -- here and by the fused AdamW, on its stream."""
        st = self._fp8
        if st is None or st.get("qt") is None or not st["weights"]:
            return
        L.check(L.load().crct_fp8_transpose_weights(st["q"].data_ptr(), st["qt"].data_ptr(), st["seg_off"].data_ptr(), st["w_out"].data_ptr(),
                                                    st["w_in"].data_ptr(), st["tile_begin"].data_ptr(), len(st["weights"]), st["n_tiles"],
                                                    int(max_workgroups), stream), "fp8_transpose_weights")

    def _fp8_step_args(self, eng):
        st = self._fp8_state(eng)
        return (st["q"], st["w_scale"], st["a_scale"], st["a_amax"])

    FP8_AMAX_WINDOW = 256      # steps between resets of the running activation / weight maxima (see crct_fp8_update_scales)

    def _fp8_update_act_scales(self):
        st = self._fp8
        st["updates"] = st.get("updates", 0) + 1
        L.check(L.load().crct_fp8_update_scales(st["a_scale"].data_ptr(), st["a_amax"].data_ptr(), st["n_sites"],
                                                int(st["updates"] % self.FP8_AMAX_WINDOW == 0), None, 448.0, L.current_stream()), "fp8_update_scales")

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self._flat_p.device))
        if probe.dtype != torch.float32:
            raise RuntimeError("CrctModel keeps fp32 master weights (bf16 compute is internal); .half()/.bfloat16() is not supported")
        if probe.device != self._flat_p.device:
            if probe.device.type != "cuda":
                raise RuntimeError("CrctModel cannot leave the GPU: the CRCT step has no CPU path")
            self._flat_p, self._flat_g, self._flat_b16 = fn(self._flat_p), fn(self._flat_g), self._flat_b16.to(probe.device)
            self._anchor = torch.zeros(1, device=probe.device, requires_grad=True)
            self._engine = None
            # everything bound to the old device goes with the engine: fp8 shadow / scales / chunk tables, segment events, the
            # lazy-clear plan, constants (the fused optimizer drops its own device state when it sees the new buffers)
            self._fp8 = None
            self._seg_done = None
            self._lazy_plan_key, self._lazy_plan = None, None
            self._const_zeros = None
            self._param_events = None
            self._opt_stream = None
            self._rebind()
        return self

    def _rebind(self):
        """After a move to another device.  The Parameters keep their identity (optimizers may already hold them), so their
        ``.data`` is re-pointed at views of the new flat buffer; such views carry their OWN version counters, so from here
        on the bf16 shadow refresh also watches the parameters' counters (``_refresh_shadow``)."""
        byname = self._params_by_name()
        for e in self.table:
            p = byname[e.name]
            p.data = self._flat_p[e.offset:e.offset + e.numel].view(e.shape)
            if e.used:
                p.grad = self._flat_g[e.offset:e.offset + e.numel].view(e.shape)
        self._rebound = list(byname.values())
        self._invalidate_shadow()

    # ------------------------------------------------------------------ flat views for the optimizer / DDP
    @property
    def flat_params(self):
        return self._flat_p

    @property
    def flat_grads(self):
        return self._flat_g

    @property
    def flat_shadow(self):
        return self._flat_b16

    def note_params_updated_natively(self):
        """Called by the fused optimizer, which rewrites the flat buffers and the shadow itself."""
        self._shadow_ver = self._param_version()

    def zero_flat_grads(self, lazy=False):
        """Clear the gradients.  ``lazy``: only the gradients that backward ACCUMULATES into (biases, LayerNorm, embeddings,
        heads: one launch over ~0.1 GB) -- the Linear weight gradients, which the engine produces with exactly one GEMM
        each, are left as they are and WRITTEN by the next backward pass (CrctStepCfg.wgrad_overwrite).  Their ``.grad``
        therefore holds stale values between ``zero_grad()`` and ``backward()``; the values after backward are bit-identical
        to the eager path.  Falls back to the full fill until one backward pass has run."""
        plan = self._lazy_zero_plan() if lazy else None
        if plan is None:
            self._flat_g.zero_()
            self._full_clears = getattr(self, "_full_clears", 0) + 1       # crct/ddp.py: NaN-filled views of owned gradients are gone
            self._wgrad_overwrite_next = False
            return
        off, num, blk_seg, blk_off, n_blk = plan
        L.check(L.load().crct_zero_runs(self._flat_g.data_ptr(), off.data_ptr(), num.data_ptr(), blk_seg.data_ptr(),
                                        blk_off.data_ptr(), n_blk, L.current_stream()), "zero_runs")
        self._wgrad_overwrite_next = True

    def non_owned_grad_runs(self):
        """Host list of (offset, numel) runs of the flat gradient buffer that backward ACCUMULATES into (everything but the Linear
        weight gradients the engine owns); None before the first engine exists.  Same runs as the lazy clear's."""
        if self._lazy_zero_plan() is None:
            return None
        return list(self._lazy_runs)

    def _lazy_zero_plan(self):
        """Device tables for crct_zero_runs over the complement of the engine's owned weight gradients.  The owned set is
        fixed by the layout (crct_engine_wgrad_owned), so one plan serves every engine this model creates; it is keyed on
        the set itself and ``_run_backward`` re-checks the key against the engine that actually runs the pass."""
        eng = self._engine
        if eng is None:
            return None
        key = eng.wgrad_owned_key()
        if self._lazy_plan_key == key:
            return self._lazy_plan
        owned = list(zip(*key)) if key[0] else []
        starts = [o for o, _ in owned]
        import bisect
        runs = []
        for e in sorted((e for e in self.table if e.used), key=lambda e: e.offset):
            k = bisect.bisect_right(starts, e.offset) - 1
            if k >= 0 and e.offset + e.numel <= owned[k][0] + owned[k][1]:      # a fused QKV weight is one owned range over three tensors
                continue
            if runs and runs[-1][0] + runs[-1][1] == e.offset:
                runs[-1][1] += e.numel
            else:
                runs.append([e.offset, e.numel])
        self._lazy_runs = [tuple(r) for r in runs]
        plan = None
        if owned and runs:
            dev = self._flat_g.device
            num_host = torch.tensor([r[1] for r in runs], dtype=torch.int64)
            lib = L.load()
            n_blk = lib.crct_adamw_plan(num_host.data_ptr(), len(runs), None, None, 0)
            blk_seg = torch.empty(n_blk, dtype=torch.int32)
            blk_off = torch.empty(n_blk, dtype=torch.int64)
            lib.crct_adamw_plan(num_host.data_ptr(), len(runs), blk_seg.data_ptr(), blk_off.data_ptr(), n_blk)
            plan = (torch.tensor([r[0] for r in runs], dtype=torch.int64, device=dev), num_host.to(dev), blk_seg.to(dev),
                    blk_off.to(dev), int(n_blk))
        self._lazy_plan_key, self._lazy_plan = key, plan
        return plan

    def _ensure_grad_views(self):
        """``optimizer.zero_grad()`` of stock torch sets ``.grad = None``: re-attach the views and clear."""
        missing = False
        byname = None
        for e in self.table:
            if not e.used:
                continue
            if byname is None:
                byname = self._params_by_name()
            if byname[e.name].grad is None:
                missing = True
                break
        if missing:
            self._flat_g.zero_()
            self._wgrad_overwrite_next = False
            for e in self.table:
                if e.used:
                    byname[e.name].grad = self._flat_g[e.offset:e.offset + e.numel].view(e.shape)

    # ------------------------------------------------------------------ engine plumbing
    def _get_engine(self, B, T, V):
        eng = self._engine
        if eng is None or B > eng.max[0] or T > eng.max[1] or V > eng.max[2]:
            mb = max(B, eng.max[0] if eng else 0)
            mt = max(T, eng.max[1] if eng else 0)
            mv = max(V, eng.max[2] if eng else 0)
            self._engine = StepEngine(self.config, self.params, mb, mt, mv, self._flat_p.device, self.cls_dropout)
            for pol in getattr(self, "site_policy", ()):          # developer / test overrides of the per-site launch policy
                self._engine.set_site_policy(**pol)
            if getattr(self, "wgrad_workgroups", None) is not None:      # (target, max rows) of crct_engine_set_wgrad_workgroups
                L.check(self._engine.lib.crct_engine_set_wgrad_workgroups(self._engine.handle, int(self.wgrad_workgroups[0]), int(self.wgrad_workgroups[1])), "set_wgrad_workgroups")
            if getattr(self, "wgrad_flush", None) is not None:    # where a layer's weight gradients leave for the side stream (crct_engine_set_wgrad_flush)
                L.check(self._engine.lib.crct_engine_set_wgrad_flush(self._engine.handle, int(self.wgrad_flush)), "set_wgrad_flush")
            mode = getattr(self, "stream_mode", None)             # (use_visual_stream, use_wgrad_streams) of crct_engine_set_streams
            if mode is not None:
                L.check(self._engine.lib.crct_engine_set_streams(self._engine.handle, int(mode[0]), int(mode[1])), "set_streams")
        return self._engine

    def aux_stream(self):
        """THE side stream of the host-side glue: the optimizer's overlapped update runs on it during the next forward, the
        data-parallel exchange (pack, collectives) during backward -- one stream, because MI355X schedules HIP streams onto
        4 hardware queues and streams that share a queue are serialised: the engine's internal streams already use them."""
        if getattr(self, "_aux_stream", None) is None or self._aux_stream.device != self._flat_p.device or self._aux_engine is not self._engine:
            if self._engine is None:
                raise RuntimeError("the auxiliary stream belongs to the step engine: run a forward pass first")
            n = C.c_int(0)
            ptr = self._engine.lib.crct_engine_aux_stream(self._engine.handle, L.current_stream(), C.byref(n))
            if not ptr:
                raise RuntimeError("crct_engine_aux_stream failed: %s" % self._engine.lib.crct_last_error().decode())
            self._aux_stream = torch.cuda.ExternalStream(ptr, device=self._flat_p.device)
            self._aux_engine, self.queue_classes = self._engine, int(n.value)
        return self._aux_stream

    def segment_done_events(self):
        """4 events per backward segment, recorded by the engine on its internal streams when the segment is enqueued."""
        if self._seg_done is None:
            from .events import DeviceEvent
            self._seg_done = [DeviceEvent() for _ in range(4 * self._engine.n_segments)]
        return self._seg_done

    def take_segment_done_events(self):
        """For the optimizer's early mode: per segment, the callables ``w(stream)`` that order ``stream`` after the
        segment's final gradients of the LAST backward pass (None if that pass did not record them)."""
        waits, self._grad_waits = self._grad_waits, None
        return waits

    def _run_backward(self, tensors, step):
        if self._opt_stream is not None:             # overlapped optimizer update / gradient memset of the previous step
            from .events import order_streams
            order_streams(self._opt_stream, torch.cuda.current_stream())
        self._ensure_grad_views()
        self._grads_dirty = True                     # gradients are being accumulated again (optimizer bookkeeping)
        eng = self._engine
        self._grad_waits = None
        if self._wgrad_overwrite_next:               # first pass since a lazy clear: the owned weight gradients hold stale values
            if self._lazy_plan_key is not None and eng.wgrad_owned_key() == self._lazy_plan_key:
                step = dict(step, wgrad_overwrite=True)     # ... and this engine writes exactly those (an engine rebuilt for a
            else:                                           # larger batch in between owns the same set: it follows from the layout)
                self._flat_g.zero_()                 # an engine with another owned set: nothing else has been accumulated yet
        self._wgrad_overwrite_next = False           # a further pass before the next clear accumulates
        self._backward_passes = getattr(self, "_backward_passes", 0) + 1
        if self.fp8_backward and self._fp8 is not None and step.get("fp8") is not None and self._fp8["n_gsites"] > 0:
            # fp8 data gradients: the first pass only collects the gradient maxima (its GEMMs run in bf16), from then on the e5m2
            # copies are quantised with the scales of the previous passes (delayed scaling, running maxima over a window)
            st = self._fp8
            mode = 1 if st["bwd_calibrated"] else 2
            if mode == 1 and not st.get("g_scales_fresh"):      # normally done by the optimizer on its stream (_fp8_update_grad_scales)
                self._fp8_update_grad_scales(L.current_stream())
            st["g_scales_fresh"] = False
            st["bwd_calibrated"] = True
            step = dict(step, fp8_bwd=(mode, st["qt"], st["g_scale"], st["g_amax"], self.fp8_wgrad))
        if self._ddp is None and self.record_segment_events:
            evs = self.segment_done_events()
            step = dict(step, seg_done_events=evs)
            eng.backward(self._flat_p, self._flat_b16, self._flat_g, tensors, step, -1)
            self._grad_waits = [[(lambda st, ev=ev: ev.wait(st)) for ev in evs[4 * i:4 * i + 4]] for i in range(eng.n_segments)]
            return
        if self._ddp is None:
            if getattr(self, "force_segmented", False):    # developer switch: the DDP call pattern without the collectives
                for i in range(eng.n_segments):
                    eng.backward(self._flat_p, self._flat_b16, self._flat_g, tensors, step, i)
                return
            eng.backward(self._flat_p, self._flat_b16, self._flat_g, tensors, step, -1)
        else:
            self._ddp.backward(self, eng, tensors, step)

    def _device_inputs(self, input_ids, txt_loc, image_feat, image_loc, token_type_ids, attention_mask,
                       image_attention_mask, image_target, R, labels):
        dev = self._flat_p.device

        def to(t, dtype):          # no kernel when the tensor is already on the device with this dtype (resident batches)
            return t.to(device=dev, dtype=dtype, non_blocking=True).contiguous()
        B, T = input_ids.shape
        V = image_feat.shape[1]
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        t = dict(tokens=to(input_ids, torch.int64), segments=to(token_type_ids, torch.int64), loc=to(txt_loc, torch.float32),
                 image_feat=to(image_feat, torch.bfloat16 if image_feat.dtype == torch.bfloat16 else torch.float32),   # bf16 features are taken as shipped
                 image_loc=to(image_loc, torch.float32),
                 image_target=to(image_target, torch.int64), R=to(R, torch.float32))
        # key masks: the engine builds them from sep_indices / hist_len (SequenceMask) and the integer image mask; any other
        # mask tensor is reduced to uint8 here
        if isinstance(attention_mask, SequenceMask):
            t["sep_indices"] = to(attention_mask.sep_indices, torch.int64)
            t["hist_len"] = to(attention_mask.hist_len.reshape(-1), torch.int64)
        elif attention_mask is None:
            t["text_keymask"] = torch.ones(B, T, dtype=torch.uint8, device=dev)
        else:
            t["text_keymask"] = to(attention_mask != 0, torch.uint8)
        if image_attention_mask is None:
            t["image_keymask"] = torch.ones(B, V, dtype=torch.uint8, device=dev)
        elif image_attention_mask.dtype == torch.int64:
            t["image_mask"] = to(image_attention_mask, torch.int64)
        else:
            t["image_keymask"] = to(image_attention_mask != 0, torch.uint8)
        if labels is not None:
            t["labels"] = to(labels.reshape(-1), torch.int64)
        return t

    # ------------------------------------------------------------------ forward (vilbert.py:1540-1661)
    def forward(self, input_ids, txt_loc, image_feat, image_loc, sep_indices=None, sep_len=None, token_type_ids=None,
                attention_mask=None, image_attention_mask=None, masked_lm_labels=None, image_label=None, image_target=None,
                next_sentence_label=None, output_all_attention_masks=False, gt_reg=None, areas=None, legend_pred=None):
        if areas is not None:
            raise NotImplementedError("'areas' belongs to the figure_qa / dvqa datasets (vilbert.py:1464-1489), out of scope")
        if gt_reg is None:
            raise ValueError("gt_reg=[R, kind] is required (vilbert.py:1586)")
        if image_target is None:
            raise ValueError("image_target is required by the image embeddings (vilbert.py:1479)")
        R, kind = gt_reg[0], gt_reg[1]
        train_branch = masked_lm_labels is not None and next_sentence_label is not None and image_target is not None
        tensors = self._device_inputs(input_ids, txt_loc, image_feat, image_loc, token_type_ids, attention_mask,
                                      image_attention_mask, image_target, R, next_sentence_label if train_branch else None)
        B, T = tensors["tokens"].shape
        V = tensors["image_feat"].shape[1]
        eng = self._get_engine(B, T, V)
        self._refresh_shadow()
        self._calls += 1
        p = self.params
        step = dict(training=self.training, use_l1=bool(p["L1"]), kind_l1=(kind == "L1"), tol_margin=float(p["tol_margin"]),
                    nsp_coeff=float(p.get("nsp_loss_coeff", 1.0)), reg_coeff=float(p.get("reg_loss_coeff", 1.0)),
                    seed=(self._seed + self._calls * 7919 + int(p.get("rank", 0)) * 104729) & 0x3FFFFFFFFFFFFFFF,
                    seg_events=self._param_events, residual_fp32=self.residual_fp32)
        self._param_events = None
        dev = self._flat_p.device
        if self.fp8 and not self.fp8_forward and not self.fp8_wgrad:
            # fp8 DATA GRADIENTS only: nothing of the forward pass is quantised (no e4m3 copies, no activation maxima); the backward
            # pass multiplies e5m2 gradients (copies written by the backward kernels themselves) with the transposed e4m3 weight shadow
            step["fp8"] = self._fp8_step_args(eng)
            step["fp8_backward_only"] = True
            if not (train_branch and torch.is_grad_enabled()):
                del step["fp8"]
        elif self.fp8:
            step["fp8"] = self._fp8_step_args(eng)
            if not self._fp8["calibrated"]:            # first fp8 forward: one dry pass collects every activation amax
                eng.forward(self._flat_p, self._flat_b16, tensors, dict(step, seg_events=None, fp8_mode=2))      # bf16 GEMMs, maxima only
                self._fp8["calibrated"] = True
            if self.training or not self._fp8.get("scaled"):
                self._fp8_update_act_scales()          # delayed scaling: this pass quantises with the previous pass's amax
                self._fp8["scaled"] = True
            if not self.fp8_forward:
                step["fp8_mode"] = 2                   # bf16 forward GEMMs; the e4m3 copies / maxima for the fp8 backward are still written
        if train_branch and torch.is_grad_enabled():
            loss, nsp, reg_loss, logits, reg, stats = _StepFn.apply(self._anchor, self, tensors, step)
        else:
            eng.forward(self._flat_p, self._flat_b16, tensors, step)
            logits, reg, stats = eng.snapshot(B)
            loss, nsp, reg_loss = stats[0], stats[1:2], reg[1]
        # the combined training loss as the head kernel computed it (differentiable); the step adapter returns it instead
        # of re-deriving it from nsp_loss / reg_loss with five more torch kernels and their autograd nodes
        self.last_loss = loss if train_branch else None
        self.loss_coeffs = (step["nsp_coeff"], step["reg_coeff"])
        self.last_stats = stats
        if self.sync_stats:
            right = (int(stats[4].item()), int(stats[5].item()))          # vilbert.py:1647 (.item() host syncs)
        else:
            right = (stats[4], stats[5])
        reg_out = [reg[0], reg_loss, reg[2], right, reg[4]]
        if self._const_zeros is None or self._const_zeros[0].device != dev:
            # the placeholder losses of vilbert.py:1583,1652-1653 (lm, image, legend): constant tensors made once, not three
            # fill kernels per step
            self._const_zeros = (torch.zeros(1, 1, device=dev), torch.zeros(1, 1, device=dev), torch.zeros(1, device=dev))
        lm_zero, img_zero, legend_loss = self._const_zeros
        if train_branch:
            return lm_zero, img_zero, nsp, None, None, logits, reg_out, legend_loss        # vilbert.py:1659
        return None, None, logits, None, None, reg_out, legend_loss                         # vilbert.py:1661


class VisualDialogEncoder(nn.Module):
    """encoder_decorator.py:9-54.  ``params['model_config']`` must exist (same assertion as :14).

    The reference builds its model with ``from_pretrained('bert-base-uncased', ...)`` (encoder_decorator.py:16), i.e. it STARTS FROM
    BERT-base for the tensors that have a counterpart there (text embeddings, the twelve text layers, the LM head) and downloads the
    archive when it is not cached.  Here the archive is a local path: ``pretrained=`` (or ``params['bert_pretrained']``) names a
    directory holding ``pytorch_model.bin``, a ``.bin`` file or a ``.tar.gz`` archive -- or is a state dict -- and is loaded with the
    reference's rules (crct/pretrained.py: gamma / beta rename, ``bert.`` prefix rule, name-and-shape matching, size mismatch =
    RuntimeError).  Without it the model keeps the ``init_bert_weights`` initialisation -- exactly the state of the reference model
    right before its weight loading (vilbert.py:1205) -- and says so once: that is a different starting point from the reference's."""

    _warned_no_pretrained = False

    def __init__(self, params, config=None, pretrained=None):
        super().__init__()
        if config is None:
            config_path = params["model_config"]
            assert os.path.exists(config_path), "model_config file not found"
            config = BertConfig.from_json_file(config_path)
        self.bert_pretrained = CrctModel(config, params=params)
        if pretrained is None:
            pretrained = params.get("bert_pretrained")
        if pretrained is not None:
            from .pretrained import load_pretrained
            self.pretrained_missing, self.pretrained_unexpected = load_pretrained(self.bert_pretrained, pretrained)
        elif not VisualDialogEncoder._warned_no_pretrained and not params.get("quiet_init", False):
            VisualDialogEncoder._warned_no_pretrained = True
            import warnings
            warnings.warn("VisualDialogEncoder: no BERT-base checkpoint given (pretrained=<path> or params['bert_pretrained']); the model starts "
                          "from init_bert_weights, whereas the reference starts its text stream from bert-base-uncased "
                          "(encoder_decorator.py:16).  Loading a CRCT checkpoint afterwards (train.py:91-130) makes this moot.")
        self.bert_pretrained.train()

    def forward(self, input_ids, txt_loc, image_feat, image_loc, sep_indices=None, sep_len=None, token_type_ids=None,
                attention_mask=None, masked_lm_labels=None, next_sentence_label=None, head_mask=None,
                random_round_indices=None, output_nsp_scores=False, output_lm_scores=False, image_attention_mask=None,
                image_label=None, image_target=None, gt_reg=None, areas=None, legend_pred=None):
        masked_lm_loss = masked_img_loss = nsp_loss = prediction_scores_t = None
        kw = dict(sep_indices=sep_indices, sep_len=sep_len, token_type_ids=token_type_ids, attention_mask=attention_mask,
                  masked_lm_labels=masked_lm_labels, next_sentence_label=next_sentence_label,
                  image_attention_mask=image_attention_mask, image_label=image_label, image_target=image_target,
                  gt_reg=gt_reg, areas=areas, legend_pred=legend_pred)
        if next_sentence_label is not None and masked_lm_labels is not None and image_target is not None:
            masked_lm_loss, masked_img_loss, nsp_loss, _, prediction_scores_t, seq_relationship_score, reg_loss, legend_loss = \
                self.bert_pretrained(input_ids, txt_loc, image_feat, image_loc, **kw)
        else:
            prediction_scores_t, _, seq_relationship_score, _, _, reg_loss, legend_loss = \
                self.bert_pretrained(input_ids, txt_loc, image_feat, image_loc, **kw)
        out = (masked_lm_loss, masked_img_loss, nsp_loss, seq_relationship_score)
        if output_lm_scores:
            out = out + (prediction_scores_t,)
        return out + (reg_loss, legend_loss)
