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
        # step outputs in ONE buffer [stats 8 | reg 5*maxB | logits 2*maxB] so that a snapshot is a single copy
        self.out = torch.zeros(8 + 7 * B, device=self.device)
        self.stats, self.reg, self.logits = self.out[:8], self.out[8:8 + 5 * B].view(5, B), self.out[8 + 5 * B:].view(B, 2)
        self._keep = None
        # hipGraph mode: kernel arguments are baked at capture, so everything that changes per step lives in
        # persistent device buffers: the dropout seed, the batch (staged copies), the upstream loss gradients
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.g_nsp_dev = torch.zeros(1, device=self.device)
        self.g_reg_dev = torch.zeros(B, device=self.device)
        self._stage = {}
        self.gstream = torch.cuda.Stream(device=self.device)     # stream capture is illegal on the default (NULL) stream

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
        b.tokens, b.segments, b.loc, b.text_keymask = L.ptr(t["tokens"]), L.ptr(t["segments"]), L.ptr(t["loc"]), L.ptr(t["text_keymask"])
        b.image_feat, b.image_loc = L.ptr(t["image_feat"]), L.ptr(t["image_loc"])
        b.image_target, b.image_keymask = L.ptr(t["image_target"]), L.ptr(t["image_keymask"])
        b.R, b.labels = L.ptr(t["R"]), L.ptr(t.get("labels"))
        b.B, b.T, b.V = t["tokens"].shape[0], t["tokens"].shape[1], t["image_feat"].shape[1]
        return b

    def stage_batch(self, tensors):
        """Copy a device batch into persistent buffers (stable addresses for graph replay)."""
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items()))
        st = self._stage.get(key)
        if st is None:
            st = {k: torch.empty_like(v) for k, v in tensors.items()}
            self._stage[key] = st
        for k, v in tensors.items():
            st[k].copy_(v, non_blocking=True)
        return st

    def _cfg(self, step):
        c = L.StepCfg()
        c.training, c.use_l1, c.kind_l1 = int(step["training"]), int(step["use_l1"]), int(step["kind_l1"])
        c.tol_margin, c.nsp_coeff, c.reg_coeff = step["tol_margin"], step["nsp_coeff"], step["reg_coeff"]
        c.grad_scale = step.get("grad_scale", 1.0)
        c.use_graph = int(bool(step.get("use_graph", False)))
        c.wgrad_overwrite = int(bool(step.get("wgrad_overwrite", False)))
        g_nsp, g_reg = step.get("g_nsp"), step.get("g_reg")
        if c.use_graph:
            if not step.get("_seed_set"):
                self.seed_dev.fill_(int(step["seed"]))
                step["_seed_set"] = True
            c.seed = (1 << 63) | self.seed_dev.data_ptr()
            if g_nsp is not None:
                self.g_nsp_dev.copy_(g_nsp.reshape(1))
                g_nsp = self.g_nsp_dev
            if g_reg is not None:
                n = g_reg.numel()
                self.g_reg_dev[:n].copy_(g_reg)
                g_reg = self.g_reg_dev
        else:
            c.seed = int(step["seed"])
        c.g_nsp_dev, c.g_reg_dev = L.ptr(g_nsp), L.ptr(g_reg)
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
        return c

    def wgrad_owned(self):
        """(offsets, numels) of the weight gradients the engine may overwrite (see CrctStepCfg.wgrad_overwrite); valid after
        at least one complete backward pass, and the set the engine honours from this call on."""
        n = self.lib.crct_engine_wgrad_owned(self.handle, None, None, 0)
        if n <= 0:
            return [], []
        off, num = (C.c_int64 * n)(), (C.c_int64 * n)()
        n2 = self.lib.crct_engine_wgrad_owned(self.handle, off, num, n)
        assert n2 == n
        return list(off), list(num)

    def _enter(self, graph):
        """Stream the engine call is enqueued on: the caller's current stream, or (graph mode) the engine's own
        capturable stream, ordered after the current stream."""
        if not graph:
            return L.current_stream()
        self.gstream.wait_stream(torch.cuda.current_stream())
        return self.gstream.cuda_stream

    def _leave(self, graph):
        if graph:
            torch.cuda.current_stream().wait_stream(self.gstream)

    def forward(self, p32, p16, tensors, step):
        B = tensors["tokens"].shape[0]
        b, c = self._batch(tensors), self._cfg(step)
        self._keep = (tensors, step)
        stream = self._enter(c.use_graph)
        L.check(self.lib.crct_engine_forward(self.handle, p32.data_ptr(), p16.data_ptr(), C.byref(b), C.byref(c),
                                             self.workspace.data_ptr(), self.logits.data_ptr(), self.reg.data_ptr(),
                                             self.stats.data_ptr(), stream), "engine_forward")
        self._leave(c.use_graph)
        return self.outputs_of(self.out, B)

    def outputs_of(self, out, B):
        """(logits [B,2], reg [5,B], stats [8]) views of an output buffer (``self.out`` or a snapshot of it)."""
        o = 8 + 5 * self.max[0]
        return out[o:o + 2 * B].view(B, 2), out[8:8 + 5 * B].view(5, B), out[:8]

    def snapshot(self, B):
        return self.outputs_of(self.out.clone(), B)

    def backward(self, p32, p16, g32, tensors, step, seg=-1):
        b, c = self._batch(tensors), self._cfg(step)
        stream = self._enter(c.use_graph)
        L.check(self.lib.crct_engine_backward(self.handle, p32.data_ptr(), p16.data_ptr(), C.byref(b), C.byref(c),
                                              self.workspace.data_ptr(), g32.data_ptr(), self.logits.data_ptr(),
                                              self.reg.data_ptr(), self.stats.data_ptr(), int(seg), stream),
                "engine_backward")
        self._leave(c.use_graph)

    def tap(self, name, B, T, V):
        n_max = B * max(T * self.cfg.hidden_size, V * self.cfg.v_hidden_size)
        out = torch.empty(n_max, dtype=torch.bfloat16, device=self.device)
        n = self.lib.crct_engine_tap(self.handle, self.workspace.data_ptr(), name.encode(), B, T, V, out.data_ptr(), n_max,
                                     L.current_stream())
        if n < 0:
            raise RuntimeError("tap(%s): %s" % (name, self.lib.crct_last_error().decode()))
        width = self.cfg.hidden_size if name.endswith(".t") or name == "seq_t" else self.cfg.v_hidden_size
        return out[:n].view(B, -1, width)
