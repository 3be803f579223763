"""Python handle of the native step engine (crct_engine_* in include/crct_hip.h).

One ``forward`` call = embeddings + 24 encoder steps + heads + loss on the current HIP stream;
``backward`` = the gradients of all 524 used tensors accumulated into the flat fp32 gradient buffer,
optionally segment by segment so the caller can launch gradient all-reduces in between.
"""
import ctypes as C

import torch

from . import lib as L
from .layout import parameter_table


def model_dims(cfg, params, cls_dropout=0.1):
    d = L.ModelDims()
    d.vocab, d.n_pos, d.n_types = cfg.vocab_size, cfg.max_position_embeddings, cfg.plotqa_vocab_types
    d.H, d.L, d.heads, d.I = cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size
    d.Fv, d.Hv, d.Lv, d.v_heads, d.Iv = (cfg.v_feature_size, cfg.v_hidden_size, cfg.v_num_hidden_layers,
                                         cfg.v_num_attention_heads, cfg.v_intermediate_size)
    d.Hb, d.b_heads, d.n_color = cfg.bi_hidden_size, cfg.bi_num_attention_heads, params["categories"] + 1
    d.n_conn = len(cfg.v_biattention_id)
    for i, (v, t) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        d.v_biatt[i], d.t_biatt[i] = v, t
    d.fusion_sum = int(cfg.fusion_method == "sum")
    d.with_coattention = int(bool(cfg.with_coattention))
    d.p_hidden, d.p_attn = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    d.p_v_hidden, d.p_v_attn = cfg.v_hidden_dropout_prob, cfg.v_attention_probs_dropout_prob
    d.p_cls = cls_dropout
    return d


class StepEngine(object):
    def __init__(self, cfg, params, max_B, max_T, max_V, device, cls_dropout=0.1):
        self.lib = L.load()
        self.cfg, self.params = cfg, params
        self.table, self.total = parameter_table(cfg, params)
        self.max = (int(max_B), int(max_T), int(max_V))
        names = "\n".join(e.name for e in self.table).encode()
        offs = (C.c_int64 * len(self.table))(*[e.offset for e in self.table])
        # size 0 marks the tensors that never receive a gradient: they are left out of the backward segments' ranges
        sizes = (C.c_int64 * len(self.table))(*[e.numel if e.used else 0 for e in self.table])
        dims = model_dims(cfg, params, cls_dropout)
        self.handle = self.lib.crct_engine_create(C.byref(dims), names, offs, sizes, len(self.table), *self.max)
        if not self.handle:
            raise RuntimeError("crct_engine_create failed: %s" % self.lib.crct_last_error().decode())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the CRCT step engine runs on an MI355X only (device=%s); there is no CPU fallback" % device)
        self.ws_bytes = self.lib.crct_engine_workspace_bytes(self.handle)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        self.n_segments = self.lib.crct_engine_num_segments(self.handle)
        self.segments = []
        lo, hi = C.c_int64(), C.c_int64()
        for s in range(self.n_segments):
            L.check(self.lib.crct_engine_segment_range(self.handle, s, C.byref(lo), C.byref(hi)), "segment_range")
            self.segments.append((lo.value, hi.value))
        B = self.max[0]
        # step outputs in ONE buffer [stats 24 | reg 5*maxB | logits 2*maxB] so that a snapshot is a single copy
        self.NS = 24                                   # 17 used: see CrctHeadArgs.stats (8 step values + the 9 floats of train.py:181)
        self.out = torch.zeros(self.NS + 7 * B, device=self.device)
        self.stats, self.reg = self.out[:self.NS], self.out[self.NS:self.NS + 5 * B].view(5, B)
        self.logits = self.out[self.NS + 5 * B:].view(B, 2)
        self._keep = None
        self._owned_key = None
        self._staged, self._op_handle = None, None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.crct_engine_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- batch marshalling: tensors must already be on the device with the dtypes of the C ABI
    def _batch(self, t):
        b = L.Batch()
        b.tokens, b.segments, b.loc = L.ptr(t["tokens"]), L.ptr(t["segments"]), L.ptr(t["loc"])
        b.image_feat, b.image_loc, b.image_target = L.ptr(t["image_feat"]), L.ptr(t["image_loc"]), L.ptr(t["image_target"])
        # key masks: ready-made uint8 masks, or what the data loader ships (the engine builds the masks itself)
        b.text_keymask, b.image_keymask = L.ptr(t.get("text_keymask")), L.ptr(t.get("image_keymask"))
        if t.get("sep_indices") is not None:
            b.sep_indices, b.hist_len, b.sep_stride = L.ptr(t["sep_indices"]), L.ptr(t["hist_len"]), t["sep_indices"].shape[1]
        b.image_mask = L.ptr(t.get("image_mask"))
        b.image_feat_bf16 = int(t["image_feat"].dtype == torch.bfloat16)
        b.R, b.labels = L.ptr(t["R"]), L.ptr(t.get("labels"))
        b.B, b.T, b.V = t["tokens"].shape[0], t["tokens"].shape[1], t["image_feat"].shape[1]
        return b

    def _cfg(self, step):
        c = L.StepCfg()
        c.training, c.use_l1, c.kind_l1 = int(step["training"]), int(step["use_l1"]), int(step["kind_l1"])
        c.tol_margin, c.nsp_coeff, c.reg_coeff = step["tol_margin"], step["nsp_coeff"], step["reg_coeff"]
        c.grad_scale = step.get("grad_scale", 1.0)
        c.wgrad_overwrite = int(bool(step.get("wgrad_overwrite", False)))
        c.grads_bf16 = L.ptr(step.get("grads_bf16")) if c.wgrad_overwrite else None
        c.seed = int(step["seed"])
        c.residual_fp32 = int(bool(step.get("residual_fp32", True)))
        c.g_nsp_dev, c.g_reg_dev, c.g_loss_dev = L.ptr(step.get("g_nsp")), L.ptr(step.get("g_reg")), L.ptr(step.get("g_loss"))
        f8 = step.get("fp8")
        if f8 is not None:          # (flat e4m3 weight shadow, weight scales, activation scales, activation amax): device tensors
            c.fp8 = int(step.get("fp8_mode", 1))
            c.params_fp8, c.fp8_w_scale, c.fp8_act_scale, c.fp8_act_amax = (L.ptr(t) for t in f8)
        f8b = step.get("fp8_bwd")
        if f8b is not None and f8 is not None:      # (mode, transposed e4m3 weight shadow, gradient scales, gradient amax)
            c.fp8_bwd = int(f8b[0])
            c.params_fp8_t, c.fp8_grad_scale, c.fp8_grad_amax = (L.ptr(t) for t in f8b[1:4])
            c.fp8_wgrad = int(bool(f8b[4])) if len(f8b) > 4 else 0
        evs = step.get("seg_events")
        if evs is not None:
            arr = (C.c_void_p * len(evs))(*[ev.cuda_event for ev in evs])
            step["_seg_events_keepalive"] = arr
            c.seg_ready_events = C.cast(arr, C.c_void_p)
        done = step.get("seg_done_events")
        if done is not None:
            arr = (C.c_void_p * len(done))(*[ev.cuda_event for ev in done])
            step["_seg_done_keepalive"] = arr
            c.seg_done_events = C.cast(arr, C.c_void_p)
        mask = step.get("seg_done_mask")
        if mask is not None:
            arr = (C.c_int32 * len(mask))(*[int(bool(m)) for m in mask])
            step["_seg_mask_keepalive"] = arr
            c.seg_done_mask = C.cast(arr, C.c_void_p)
        cb = step.get("seg_enqueued")
        if cb is not None:          # python callable(seg): wrapped once per call; exceptions are kept and re-raised after the engine call
            err = step.setdefault("_seg_enqueued_errors", [])

            def tramp(seg, _user, cb=cb, err=err):
                try:
                    cb(int(seg))
                except BaseException as ex:       # noqa: BLE001 -- must not unwind through the C frame
                    err.append(ex)
            fn = L.SEG_ENQUEUED_FN(tramp)
            step["_seg_enqueued_keepalive"] = fn
            c.seg_enqueued = C.cast(fn, C.c_void_p)
        return c

    def set_site_policy(self, site, kind, cfg=-1, split_k=0, phase=-1):
        """Launch policy of one GEMM site (crct_engine_set_site_policy): ``site`` / ``kind`` by name (crct.lib.SITE_NAMES /
        KIND_NAMES) or number, ``phase`` 0 = text-only part of the schedule, 1 = beside the visual stream, -1 = both."""
        s = L.SITE_NAMES.index(site) if isinstance(site, str) else int(site)
        k = L.KIND_NAMES.index(kind) if isinstance(kind, str) else int(kind)
        L.check(self.lib.crct_engine_set_site_policy(self.handle, s, k, int(phase), int(cfg), int(split_k)), "set_site_policy")

    def fp8_layout(self):
        """(number of activation scale sites, [(flat offset, numel)] of the weights with an e4m3 shadow; index = scale slot)."""
        n = self.lib.crct_engine_fp8_weights(self.handle, None, None, 0)
        off, num = (C.c_int64 * max(n, 1))(), (C.c_int64 * max(n, 1))()
        if n > 0:
            assert self.lib.crct_engine_fp8_weights(self.handle, off, num, n) == n
        return self.lib.crct_engine_fp8_sites(self.handle), list(zip(off[:n], num[:n]))

    def wgrad_owned(self):
        """(offsets, numels) of the weight gradients the engine overwrites under CrctStepCfg.wgrad_overwrite: fixed by the
        layout at engine creation (every Linear weight produced by exactly one weight-gradient GEMM per pass)."""
        return [list(x) for x in self.wgrad_owned_key()]

    def wgrad_owned_key(self):
        if self._owned_key is None:
            n = self.lib.crct_engine_wgrad_owned(self.handle, None, None, 0)
            off, num = (C.c_int64 * max(n, 1))(), (C.c_int64 * max(n, 1))()
            if n > 0:
                assert self.lib.crct_engine_wgrad_owned(self.handle, off, num, n) == n
            self._owned_key = (tuple(off[:n]), tuple(num[:n]))
        return self._owned_key

    # -- the two engine calls go through the registered torch ops (torch.ops.crct.step_forward / step_backward, crct/torch_ops.py);
    #    the step configuration holds scalars, event handles and fp8 state, so it is staged here rather than passed as tensors
    def staged_step(self):
        return self._staged

    def forward(self, p32, p16, tensors, step):
        from . import torch_ops as T
        if self._op_handle is None:
            self._op_handle = T.engine_handle(self)
        self._staged = step
        torch.ops.crct.step_forward(self._op_handle, p32, p16, T.pack_batch(tensors))
        return self.outputs_of(self.out, tensors["tokens"].shape[0])

    def backward(self, p32, p16, g32, tensors, step, seg=-1):
        from . import torch_ops as T
        if self._op_handle is None:
            self._op_handle = T.engine_handle(self)
        self._staged = step
        torch.ops.crct.step_backward(self._op_handle, p32, p16, g32, T.pack_batch(tensors), int(seg))

    def forward_native(self, p32, p16, tensors, step):
        B = tensors["tokens"].shape[0]
        b, c = self._batch(tensors), self._cfg(step)
        self._keep = (tensors, step)
        L.check(self.lib.crct_engine_forward(self.handle, p32.data_ptr(), p16.data_ptr(), C.byref(b), C.byref(c),
                                             self.workspace.data_ptr(), self.logits.data_ptr(), self.reg.data_ptr(),
                                             self.stats.data_ptr(), L.current_stream()), "engine_forward")
        return self.outputs_of(self.out, B)

    def outputs_of(self, out, B):
        """(logits [B,2], reg [5,B], stats [24]) views of an output buffer (``self.out`` or a snapshot of it)."""
        o = self.NS + 5 * self.max[0]
        return out[o:o + 2 * B].view(B, 2), out[self.NS:self.NS + 5 * B].view(5, B), out[:self.NS]

    def snapshot(self, B):
        """The step outputs as tensors of their own (ONE copy kernel): the engine rewrites its buffer on the next call."""
        return self.outputs_of(self.out.clone(), B)

    def backward_native(self, p32, p16, g32, tensors, step, seg=-1):
        b, c = self._batch(tensors), self._cfg(step)
        L.check(self.lib.crct_engine_backward(self.handle, p32.data_ptr(), p16.data_ptr(), C.byref(b), C.byref(c),
                                              self.workspace.data_ptr(), g32.data_ptr(), self.logits.data_ptr(),
                                              self.reg.data_ptr(), self.stats.data_ptr(), int(seg), L.current_stream()),
                "engine_backward")
        errs = step.get("_seg_enqueued_errors")
        if errs:
            raise errs[0]

    def tap(self, name, B, T, V):
        n_max = B * max(T * self.cfg.hidden_size, V * self.cfg.v_hidden_size)
        out = torch.empty(n_max, dtype=torch.bfloat16, device=self.device)
        n = self.lib.crct_engine_tap(self.handle, self.workspace.data_ptr(), name.encode(), B, T, V, out.data_ptr(), n_max,
                                     L.current_stream())
        if n < 0:
            raise RuntimeError("tap(%s): %s" % (name, self.lib.crct_last_error().decode()))
        width = self.cfg.hidden_size if name.endswith(".t") or name == "seq_t" else self.cfg.v_hidden_size
        return out[:n].view(B, -1, width)
