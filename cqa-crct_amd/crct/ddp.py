"""Data-parallel gradient exchange for the CRCT step: RCCL all-reduce of the flat gradient buffer,
launched WHILE backward is being enqueued and overlapped with it.

Reference behaviour being replaced: ``DistributedDataParallel(model, device_ids=[gpu],
find_unused_parameters=True)`` (CRCT/train.py:138-143) = parameter broadcast from rank 0 at
construction + bucketed gradient all-reduce (AVG) triggered by autograd hooks as backward produces
the gradients, with the 36 never-used tensors tolerated; plus the per-iteration 9-float stats
all-reduce (train.py:181-189).  Here:

  * parameters live in one flat buffer ordered by first use, so each backward segment of the native
    engine (heads, then the encoder schedule reversed, then the embeddings) completes one CONTIGUOUS
    range of the flat gradient buffer; consecutive segments are merged into buckets of >= ``bucket_mb``;
  * backward is ONE engine call.  The engine records four events (one per internal stream) after it
    has enqueued a segment and then calls back into ``FlatGradDDP`` (``CrctStepCfg.seg_enqueued``) --
    still inside the call, while the host goes on enqueuing the rest of backward -- and the bucket that
    segment completes is launched right there: the communication stream waits for the four events,
    packs the bucket to bf16 and issues ``all_reduce(SUM, async)``.  The first collective is in RCCL's
    queue ~0.3 ms into backward, as with torch DDP's autograd hooks, not after the host has enqueued
    all of it (round 2);
  * payload: ``grad_dtype=torch.bfloat16`` (default) sends 2 bytes per parameter -- 477 MB per step
    instead of 953 MB: at ~350 GB/s of bus bandwidth over xGMI that is ~2.4 ms, which
    hides behind the 4 ms backward, where the fp32 payload (4.8 ms) cannot.  What ``.grad`` holds afterwards
    (``materialize_grads``): by default (None) the whole reduced bucket is written back into the fp32 ``.grad``
    views -- DistributedDataParallel's contract, whatever reads them (a stock optimizer, clip_grad_norm_,
    GradScaler's inf check, logging) -- UNLESS this package's FusedAdamW is attached to the model: it consumes
    the all-reduced bf16 bucket as it lies (``crct_adamw_step(g_bf16=...)``), so only the gradients backward
    accumulates in fp32 (biases, LayerNorm, embeddings, heads: 5 % of the elements) are written back and the
    fp32 views of the Linear weight gradients, which then exist in the bf16 buffer only, are filled with NaN:
    a reader gets the reduced value or a NaN, never a local or stale gradient.  True / False force either.
    ``grad_dtype=torch.float32`` is the reference's payload;
  * the 1/world averaging is folded into the loss-gradient seeds (no extra pass over the gradients);
  * tensors that never receive a gradient sit at the tail of the layout and are never sent;
  * the 9-float stats all-reduce runs asynchronously on the communication stream (``AsyncStats``) and
    is waited for at the end of the step, not between forward and backward.

xGMI note (MI355X: 7 links x ~153 GB/s per GPU, point-to-point): few large messages let RCCL use all
links; 64 MB buckets of fp32 gradients (32 MB of bf16 payload) give ~15 collectives per step.
"""
import contextlib

import torch
import torch.distributed as dist


def plan_buckets(segments, bucket_elems):
    """Merge backward-ordered segments [(lo, hi)] into buckets.
    Returns [(last_segment_index, lo, hi)] -- the bucket is ready after that segment ran."""
    buckets, cur_lo, cur_hi, acc = [], None, None, 0
    for i, (lo, hi) in enumerate(segments):
        if hi <= lo:
            continue
        cur_lo = lo if cur_lo is None else min(cur_lo, lo)
        cur_hi = hi if cur_hi is None else max(cur_hi, hi)
        acc += hi - lo
        if acc >= bucket_elems:
            buckets.append((i, cur_lo, cur_hi))
            cur_lo = cur_hi = None
            acc = 0
    if cur_lo is not None:
        buckets.append((len(segments) - 1, cur_lo, cur_hi))
    return buckets


def reduce_while_running(flat_grads, segments, buckets, run_segment, group=None):
    """Run backward segment by segment; launch the async all-reduce of every finished bucket.
    ``run_segment(i)`` enqueues segment i on the current stream.  Returns after all collectives
    have been ordered before further work on the current stream (no host block on GPU).
    (The pre-event call pattern, kept as ``FlatGradDDP.event_mode = False``.)"""
    works, b = [], 0
    for i in range(len(segments)):
        run_segment(i)
        while b < len(buckets) and buckets[b][0] == i:
            _, lo, hi = buckets[b]
            works.append(dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
            b += 1
    for w in works:
        w.wait()


class BucketExchange(object):
    """The exchange of ONE backward pass: ``launch(b)`` is called (from the engine's callback) when the last segment of
    bucket b has been enqueued; ``finish()`` after the engine call launches whatever is left and orders the consumer.

    Device-agnostic: on CUDA tensors ``wait_events(b)`` makes the communication stream wait for the segment's events and
    pack / unpack are HIP kernels; the CPU / gloo tests pass plain functions."""

    def __init__(self, flat_grads, buckets, group, comm_buf=None, materialize=False, stream_ctx=None, wait_events=None,
                 pack=None, unpack=None, after_bucket=None, collective=None, pack_bucket=None, unpack_bucket=None):
        self.flat, self.buckets, self.group = flat_grads, buckets, group
        self.comm_buf, self.materialize = comm_buf, materialize
        self.stream_ctx = stream_ctx or contextlib.nullcontext
        self.wait_events = wait_events or (lambda b: None)
        self.pack = pack or (lambda src, dst: dst.copy_(src))
        self.pack_bucket = pack_bucket      # pack_bucket(b): packs only what is not in the communication buffer yet (see FlatGradDDP.backward)
        self.unpack_bucket = unpack_bucket  # unpack_bucket(b): without `materialize`, still writes back what backward accumulates in fp32
        self.unpack = unpack or (lambda src, dst: dst.copy_(src))
        self.after_bucket = after_bucket or (lambda b: None)
        # the collective: by default torch.distributed (gloo in the CPU tests; blocks until done), on the GPU RCCL called
        # directly on the communication stream (crct/rccl.py)
        self.collective = collective or (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait())
        self.works = [None] * len(buckets)
        self.issue_order = []
        self.debug_skip = ()         # timing experiments only (tools/): leave out "pack" / "collective"

    def launch(self, b):
        if self.works[b] is not None:
            return
        _, lo, hi = self.buckets[b]
        with self.stream_ctx():
            self.wait_events(b)
            if self.comm_buf is not None:
                if "pack" not in self.debug_skip:
                    if self.pack_bucket is not None:
                        self.pack_bucket(b)
                    else:
                        self.pack(self.flat[lo:hi], self.comm_buf[lo:hi])
                payload = self.comm_buf[lo:hi]
            else:
                payload = self.flat[lo:hi]
            if "collective" in self.debug_skip:
                self.works[b] = True
                self.issue_order.append(b)
                self.after_bucket(b)
                return
            self.collective(payload)
            self.works[b] = True
            self.issue_order.append(b)
            if self.comm_buf is not None and self.materialize:
                self.unpack(self.comm_buf[lo:hi], self.flat[lo:hi])
            elif self.comm_buf is not None and self.unpack_bucket is not None:
                self.unpack_bucket(b)
            self.after_bucket(b)

    def finish(self):
        for b in range(len(self.buckets)):
            self.launch(b)


class FlatGradDDP(torch.nn.Module):
    """``torch.nn.parallel.DistributedDataParallel``'s place in CRCT/train.py:138-143 and CRCT/evaluation.py:56-61, after
    ``dist.init_process_group(backend='nccl', ...)`` (RCCL on ROCm)::

        crct_model = FlatGradDDP(crct_model, device_ids=[gpu], find_unused_parameters=True)

    The result is an ``nn.Module`` WRAPPER with the surface those loops use: calling it calls the wrapped model (``forward(crct_model,
    batch, params)`` of the step adapter, train.py:173), ``.module`` is the wrapped model (``crct_model.module.state_dict()``,
    train.py:289), ``train()`` / ``eval()`` / ``to(device)`` / ``parameters()`` pass through as for any module, its own
    ``state_dict()`` carries DistributedDataParallel's ``module.`` prefix, ``no_sync()`` skips the exchange of a micro-step.  It also
    attaches itself to the model's flat-buffer core, so a loop that keeps calling the ORIGINAL model object gets the exchange too
    (bench.py does).  ``device_ids`` / ``output_device`` must name the device the model is on; ``find_unused_parameters`` is accepted
    and moot (the 36 never-used tensors sit outside the exchanged range by construction); other DistributedDataParallel keywords
    (``broadcast_buffers``, ``gradient_as_bucket_view``, ``static_graph``, ``bucket_cap_mb`` = ``bucket_mb`` ...) are accepted."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, broadcast_buffers=True, process_group=None, bucket_cap_mb=None,
                 find_unused_parameters=True, check_reduction=False, gradient_as_bucket_view=False, static_graph=False,
                 bucket_mb=64, broadcast=True, grad_dtype=torch.bfloat16, materialize_grads=None):
        super().__init__()
        from .optim import _crct_core
        self.module = module
        # NOT a child module: the core already is one of `module` (state_dict / parameters() must list every tensor once)
        self.__dict__["core"] = _crct_core(module)
        if not hasattr(self.core, "flat_params"):
            raise TypeError("FlatGradDDP wraps this package's VisualDialogEncoder / CrctModel (got %s)" % type(module).__name__)
        dev = self.core.flat_params.device
        for name, ids in (("device_ids", device_ids), ("output_device", None if output_device is None else [output_device])):
            if ids is None:
                continue
            if len(ids) != 1:
                raise ValueError("FlatGradDDP: one process per GPU (train.py:356-363): %s must name exactly one device, got %r" % (name, ids))
            want = torch.device("cuda", ids[0]) if isinstance(ids[0], int) else torch.device(ids[0])
            if dev.type == "cuda" and (want.type != "cuda" or (want.index if want.index is not None else torch.cuda.current_device()) != dev.index):
                raise ValueError("FlatGradDDP: %s=%r but the model is on %s" % (name, ids, dev))
        if bucket_cap_mb is not None:
            bucket_mb = bucket_cap_mb
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        if grad_dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("grad_dtype must be torch.bfloat16 or torch.float32")
        self.grad_dtype = grad_dtype
        self.materialize_grads = None if materialize_grads is None else bool(materialize_grads)      # None: see materializes()
        self._poisoned_at = None     # full-clear count of the gradient buffer at which the owned fp32 views were last filled with NaN
        self._buckets = None
        self._events = self._comm = self._bucket_done = None
        self._comm_buf = None
        self._seg_bucket = None
        self.event_mode = True       # False: segment-by-segment engine calls with the collectives launched in between
        self.require_sync = True
        self.force_exchange = False  # run the bucketed exchange even on a single rank (single-rank RCCL communicator: tests, probes)
        # one-GPU prediction of an N-rank run (bench.py --ghost-ranks N): behind every bucket's (single-rank) all-reduce a stand-in kernel
        # with RCCL's footprint -- `channels` workgroups streaming the bucket 2 (N - 1) / N times through HBM for as long as the ring
        # would need at `bus_GBps` -- runs on the exchange's stream (crct_ghost_collective).  dict(ranks, channels, bus_GBps) or None.
        self.ghost = None
        self.ghost_us = 0.0          # stand-in time issued in the last pass (sum over the buckets)
        self.last_exchange = None    # BucketExchange of the last synchronised backward pass
        self.issued_inside_engine_call = 0      # collectives launched from the engine's callback during the last pass
        self._grad_source_valid = False
        self._materialized_last = False
        self._rccl = None                    # None: not created yet; False: creation failed somewhere, torch.distributed carries the collectives
        self.rccl_fallback = None            # why (communicator())
        self.direct_bf16_wgrad = True       # bf16 payload: the engine writes the Linear weight gradients into the communication buffer itself
        self.packed_runs_only = False
        if broadcast:      # DDP constructor semantics: rank 0's parameters win (train.py:139)
            dist.broadcast(self.core.flat_params, 0, group=process_group)
            self.core._invalidate_shadow()
        # plain attribute of the core, NOT a registered child: core -> wrapper -> module -> core would be a cycle in the module tree
        object.__setattr__(self.core, "_ddp", self)

    def forward(self, *args, **kwargs):
        """The wrapped model's call (encoder_decorator.py:19-54 signature); the exchange itself hangs on the core's backward."""
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (``batch_multiply`` > 1): skip the exchange on all but the last micro-step."""
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    def materializes(self):
        """Does a bf16 exchange write the reduced bucket back into the fp32 ``.grad`` views?  ``materialize_grads`` if it was
        given; otherwise yes, unless this package's FusedAdamW is attached to the model (it reads the bf16 buffer itself:
        ``grad_source()``).  Decided per pass: the optimizer may be built before or after this object (train.py:85 / :139)."""
        if self.materialize_grads is not None:
            return self.materialize_grads
        # the optimizer that actually steps: FusedAdamW registers itself as a weak reference (an optimizer that was built and
        # thrown away does not count) and only one that updates EVERY gradient-receiving tensor may leave the fp32 views unwritten
        ref = getattr(self.core, "_fused_optimizer", None)
        opt = ref() if callable(ref) else None
        return not (opt is not None and opt.covers_every_gradient())

    def communicator(self):
        """The RCCL communicator of the exchange (crct/rccl.py), created on first use; None when the process group is not an RCCL
        one (the gloo-on-one-GPU tests: the collectives then go through torch.distributed)."""
        if self._rccl is None and self.core.flat_grads.is_cuda and dist.get_backend(self.group) == "nccl":
            from .rccl import Communicator
            comm, err = None, None
            try:
                comm = Communicator(self.core.flat_grads.device, self.group)
            except Exception as e:              # a failed bootstrap must not take a multi-GPU job down: torch's own RCCL group still works
                err = e
            # every rank takes the same route: one rank on the direct communicator and its peers on torch's group would wait for ever
            ok = torch.tensor([0 if comm is None else 1], device=self.core.flat_grads.device, dtype=torch.int32)
            if dist.get_world_size(self.group) > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 1:
                self._rccl = comm
            else:
                if comm is not None:
                    comm.destroy()
                self._rccl = False
                self.rccl_fallback = repr(err) if err is not None else "a peer rank could not create its communicator"
                import warnings
                warnings.warn("crct.ddp: direct RCCL communicator unavailable (%s); the gradient exchange falls back to torch.distributed's "
                              "RCCL process group (collectives on a stream of torch's choosing: slower, same results)" % self.rccl_fallback)
        return self._rccl or None

    # ------------------------------------------------------------------ what the optimizer reads
    def grad_source(self):
        """The bf16 buffer (element offsets of the flat gradient buffer) that holds the all-reduced gradients of the last
        synchronised backward pass, or None when the fp32 gradient buffer does (fp32 payload, or ``materialize_grads``)."""
        if self._grad_source_valid and self._comm_buf is not None and not self._materialized_last:
            return self._comm_buf
        return None

    def wait_all(self, stream):
        """Order ``stream`` behind every collective of the last pass (system scope: peers wrote these bytes)."""
        if self._bucket_done and self.last_exchange is not None and self.core.flat_grads.is_cuda:
            stream.wait_event(self._bucket_done[-1])       # the communication stream runs the buckets in order

    def segment_waits(self):
        """Per backward segment: callables ``w(stream)`` that order ``stream`` behind the collective of the bucket the
        segment belongs to (the optimizer's early mode: AdamW of a bucket behind THAT bucket's all-reduce)."""
        if not self._bucket_done or self._seg_bucket is None:
            return None
        return [[(lambda st, ev=self._bucket_done[b]: st.wait_event(ev))] for b in self._seg_bucket]

    # ------------------------------------------------------------------ backward
    def _pack_plans(self, core, eng):
        """Per bucket: device tables (run offsets, run lengths, chunk table) of the gradient elements the engine does NOT own, i.e.
        what still has to be cast into the bf16 communication buffer when the weight-gradient GEMMs write there themselves."""
        key = eng.wgrad_owned_key()
        if getattr(self, "_pack_plans_key", None) == key:
            return self._pack_plans_cache
        runs = core.non_owned_grad_runs()
        plans = None
        if runs is not None:
            from . import lib as L
            lib = L.load()
            dev = core.flat_grads.device
            plans = []
            for _, lo, hi in self._buckets:
                mine = [(max(o, lo), min(o + n, hi) - max(o, lo)) for o, n in runs if o < hi and o + n > lo]
                if not mine:
                    plans.append((None, None, None, None, 0))
                    continue
                num_host = torch.tensor([n for _, n in mine], dtype=torch.int64)
                n_blk = lib.crct_adamw_plan(num_host.data_ptr(), len(mine), None, None, 0)
                blk_seg, blk_off = torch.empty(n_blk, dtype=torch.int32), torch.empty(n_blk, dtype=torch.int64)
                lib.crct_adamw_plan(num_host.data_ptr(), len(mine), blk_seg.data_ptr(), blk_off.data_ptr(), n_blk)
                plans.append((torch.tensor([o for o, _ in mine], dtype=torch.int64, device=dev), num_host.to(dev), blk_seg.to(dev),
                              blk_off.to(dev), int(n_blk)))
        self._pack_plans_key, self._pack_plans_cache = key, plans
        return plans

    def _plan(self, eng):
        self._buckets = plan_buckets(eng.segments, self.bucket_elems)
        self._seg_bucket, b = [], 0
        for i in range(len(eng.segments)):
            while b < len(self._buckets) - 1 and self._buckets[b][0] < i:
                b += 1
            self._seg_bucket.append(b)
        self._last_of = {last: b for b, (last, _, _) in enumerate(self._buckets)}

    def _place_wgrad_streams(self, core, eng, exchange):
        """With an exchange in the pass the weight gradients of both data streams share ONE side stream, so that the collectives
        (auxiliary stream) have a hardware queue to themselves: the engine puts the auxiliary stream and the visual weight-gradient
        stream in the same queue class (engine.cpp, ensure_streams), and an RCCL kernel that waits for its peers must not sit in
        front of GEMMs.  Passes without an exchange (``no_sync()``, one rank) get the two weight-gradient streams back.  An explicit
        ``core.stream_mode`` (bench.py, developer A/B runs) is the caller's choice and stands."""
        if getattr(core, "stream_mode", None) is not None or not core.flat_grads.is_cuda or not hasattr(eng, "handle"):
            return
        want = 2 if exchange else 1
        if getattr(eng, "_ddp_wgrad_streams", None) != want:
            from . import lib as L
            L.check(eng.lib.crct_engine_set_streams(eng.handle, 1, want), "set_streams")
            eng._ddp_wgrad_streams = want

    def backward(self, core, eng, tensors, step):
        step = dict(step)
        if self.world > 1:      # also on accumulation-only micro-steps: the final SUM then yields the average
            step["grad_scale"] = step.get("grad_scale", 1.0) / self.world      # folded into the head kernel's gradient seeds
        self._grad_source_valid = False
        if not self.require_sync or (self.world == 1 and not self.force_exchange):
            self.last_exchange = None
            self._poisoned_at = None      # this pass writes LOCAL fp32 gradients into the owned views: a later exchange must refill the NaNs
            self._place_wgrad_streams(core, eng, False)
            eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, -1)
            return
        if self._buckets is None:
            self._plan(eng)
        self._place_wgrad_streams(core, eng, True)
        if not (self.event_mode and core.flat_grads.is_cuda):
            self.last_exchange = None
            reduce_while_running(core.flat_grads, eng.segments, self._buckets,
                                 lambda i: eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, i),
                                 self.group)
            return
        from . import lib as L
        lib = L.load()
        dev = core.flat_grads.device
        self._comm = core.aux_stream()           # the engine's auxiliary stream (the optimizer's too): idle during backward, on a hardware queue of its own
        if self._events is None:
            from .events import DeviceEvent
            # The 4 x 26 per-segment events only order the communication stream behind the engine's internal streams -- ONE
            # device -- so they are device-scope events (no system-scope fence in the record: 104 cache write-back /
            # invalidations per backward pass cost 0.9 ms of GPU time, tools/step_phases.py --exchange).  They make the
            # producers' stores (GEMM epilogues, pack kernel: plain or non-temporal stores, written through this device's L2 at
            # the end of each kernel) visible to later kernels of THIS device, which is all the collective's own kernel needs to
            # read them: ncclAllReduce is enqueued on the auxiliary stream itself (crct/rccl.py; no ProcessGroupNCCL stream or
            # event is involved any more), it reads the send buffer as an ordinary kernel of this device behind those waits, and
            # what it hands to PEERS travels through RCCL's own protocol (its fine-grained staging buffers and flags carry the
            # system-scope release / acquire).  In the other direction, what peers have written into this rank's buffer is made
            # visible by that same protocol before the RCCL kernel ends; consumers on other streams wait for `bucket_done`, a
            # stock torch event (system-scope fence in its record) recorded behind each bucket.
            self._events = [DeviceEvent() for _ in range(4 * len(eng.segments))]
            self._bucket_done = [torch.cuda.Event() for _ in self._buckets]
            for ev in self._events + self._bucket_done:          # created / recorded once so that a wait before the first record is legal
                ev.record()
        if self.grad_dtype == torch.bfloat16 and self._comm_buf is None:
            self._comm_buf = torch.zeros(core.flat_grads.numel(), dtype=torch.bfloat16, device=dev)
        comm = self._comm

        def wait_events(b):
            last = self._buckets[b][0]
            for ev in self._events[4 * last:4 * last + 4]:
                ev.wait(comm)

        def pack(src, dst):
            L.check(lib.crct_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), comm.cuda_stream), "pack gradients")

        def unpack(src, dst):
            L.check(lib.crct_cast_bf16_f32(src.data_ptr(), dst.data_ptr(), src.numel(), comm.cuda_stream), "unpack gradients")

        # The engine writes the OWNED weight gradients (every Linear weight: ~95 % of the elements) straight into the communication
        # buffer, rounded to bf16 by the GEMM epilogue (CrctStepCfg.grads_bf16): packing then only casts the runs of elements
        # backward accumulates in fp32 (biases, LayerNorm, embeddings, heads).  Only on a pass that WRITES those gradients
        # (wgrad_overwrite: the first pass after a clear -- an accumulation pass adds in fp32 and packs everything).
        materialize = self.materializes()
        direct = self.grad_dtype == torch.bfloat16 and bool(step.get("wgrad_overwrite")) and self.direct_bf16_wgrad
        plans = self._pack_plans(core, eng) if (direct or (self.grad_dtype == torch.bfloat16 and not materialize)) else None
        pack_bucket = unpack_bucket = None
        if plans is not None and direct:
            step["grads_bf16"] = self._comm_buf

            def pack_bucket(b):
                off, num, blk_seg, blk_off, n_blk = plans[b]
                if n_blk:
                    L.check(lib.crct_cast_runs_f32_bf16(core.flat_grads.data_ptr(), self._comm_buf.data_ptr(), off.data_ptr(), num.data_ptr(),
                                                        blk_seg.data_ptr(), blk_off.data_ptr(), n_blk, comm.cuda_stream), "pack gradients (runs)")
        if plans is not None and not materialize:
            # the fused optimizer reads the bf16 buffer; the gradients backward accumulates in fp32 (5 % of the elements) still go
            # back into their .grad views, and the fp32 views of the OWNED Linear weight gradients -- not written by a direct pass,
            # local values after a packed one -- are NaN from now on: readers get the reduced gradient or a NaN
            def unpack_bucket(b):
                off, num, blk_seg, blk_off, n_blk = plans[b]
                if n_blk:
                    L.check(lib.crct_cast_runs_bf16_f32(self._comm_buf.data_ptr(), core.flat_grads.data_ptr(), off.data_ptr(), num.data_ptr(),
                                                        blk_seg.data_ptr(), blk_off.data_ptr(), n_blk, comm.cuda_stream), "unpack gradients (runs)")
        self.packed_runs_only = plans is not None and direct
        rccl = self.communicator()
        collective = (lambda t: rccl.all_reduce_(t, comm)) if rccl is not None else None        # on the auxiliary stream itself: no hidden stream
        if self.ghost and self.world == 1:
            real, gh = collective, self.ghost
            self.ghost_us = 0.0

            def collective(t, real=real, gh=gh):
                if real is not None:
                    real(t)
                n = int(gh["ranks"])
                nbytes = t.numel() * t.element_size()
                us = nbytes * 2.0 * (n - 1) / n / (float(gh["bus_GBps"]) * 1e3)          # ring all-reduce: 2 (N - 1) / N of the payload per link direction
                self.ghost_us += us
                L.check(lib.crct_ghost_collective(t.data_ptr(), nbytes // 16 * 16, int(gh["channels"]), 2, us, comm.cuda_stream), "ghost_collective")
        ex = BucketExchange(core.flat_grads, self._buckets, self.group,
                            comm_buf=self._comm_buf if self.grad_dtype == torch.bfloat16 else None,
                            materialize=materialize, stream_ctx=lambda: torch.cuda.stream(comm), wait_events=wait_events,
                            pack=pack, unpack=unpack, after_bucket=lambda b: self._bucket_done[b].record(comm), collective=collective,
                            pack_bucket=pack_bucket, unpack_bucket=unpack_bucket)
        self.last_exchange = ex
        ex.debug_skip = getattr(self, "debug_skip", ())

        def on_segment(seg):          # engine callback: segment `seg` (and its four events) is enqueued
            b = self._last_of.get(seg)
            if b is not None:
                ex.launch(b)

        step["seg_done_events"] = self._events
        step["seg_enqueued"] = on_segment
        step["seg_done_mask"] = [int(i in self._last_of) for i in range(len(eng.segments))]      # events / callbacks at bucket ends only
        eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, -1)
        self.issued_inside_engine_call = len(ex.issue_order)
        ex.finish()                   # nothing left unless the engine ran without the callback
        if rccl is not None:
            rccl.check_async()        # an enqueue that succeeded can still have failed since (peer gone, transport error): fail HERE, not in a hang
        if unpack_bucket is not None:
            self._poison_owned(core, eng, direct)
        self._grad_source_valid = True
        self._materialized_last = materialize or self.grad_dtype != torch.bfloat16
        # consumers on the caller's stream are ordered behind the last bucket (collective + write-back): with `materialize` every
        # .grad view holds the reduced gradient from here on (a stock optimizer, clip_grad_norm_, a GradScaler); without it
        # (FusedAdamW attached) the views backward accumulates into do, and the Linear weights' read NaN -- the fused optimizer
        # takes those from the bf16 buffer (grad_source) and in overlap mode orders its own stream itself (wait_all /
        # segment_waits), so this wait then costs nothing extra
        torch.cuda.current_stream().wait_event(self._bucket_done[-1])

    def _poison_owned(self, core, eng, direct):
        """NaN into the fp32 views of the owned Linear weight gradients (the reduced values live in the bf16 buffer only).  After a
        direct pass nothing writes those views, so once per full clear of the gradient buffer is enough; a pass that packed from
        fp32 (accumulation, ``direct_bf16_wgrad = False``) has just written local values there, so it is repeated."""
        epoch = getattr(core, "_full_clears", 0)
        if direct and self._poisoned_at == epoch:
            return
        offs, nums = eng.wgrad_owned_key()
        with torch.cuda.stream(self._comm):          # behind every bucket's pack kernel: the exchange has read what it needed
            for o, n in zip(offs, nums):
                core.flat_grads[o:o + n].fill_(float("nan"))
            self._bucket_done[-1].record(self._comm)
        self._poisoned_at = epoch if direct else None


class AsyncStats(object):
    """train.py:181-189 without the host syncs and without stalling the step: the nine training statistics are copied into
    this object's buffer and all-reduced (SUM, asynchronously: the process group's own stream picks the work up behind the
    current stream) right after forward; ``result()`` -- at the end of the step -- orders the current stream behind the
    collective and returns the tensor with the first six entries averaged.  No stream of its own (hardware queues are scarce:
    see ``CrctModel.aux_stream``).  One instance per training loop (the buffer is reused)."""

    def __init__(self, world_size, group=None, device=None, ddp=None):
        self.world, self.group, self.ddp = world_size, group, ddp
        self.buf = torch.zeros(9, device=device)
        self.work = self.done = None

    def launch(self, stats9):
        rccl = self.ddp.communicator() if self.ddp is not None else None
        if rccl is not None:          # on the engine's auxiliary stream, behind the forward pass that produced the statistics
            from .events import order_streams
            aux = self.ddp.core.aux_stream()
            order_streams(torch.cuda.current_stream(), aux)
            with torch.cuda.stream(aux):
                self.buf.copy_(stats9)
                rccl.all_reduce_(self.buf, aux)
                if self.done is None:
                    self.done = torch.cuda.Event()
                self.done.record(aux)
            self.work = True
            return
        self.buf.copy_(stats9)
        self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def result(self):
        if self.work is None:
            return None
        if self.work is True:
            torch.cuda.current_stream().wait_event(self.done)
        else:
            self.work.wait()          # RCCL group of torch: a stream-level wait on the current stream; gloo: blocks the host
        self.work = None
        out = self.buf.clone()
        out[:-3] = out[:-3] / self.world
        return out


def all_reduce_stats(stats9, world_size, group=None):
    """train.py:181-189: SUM a 9-float stats tensor, first six entries averaged (blocking form)."""
    dist.all_reduce(stats9, op=dist.ReduceOp.SUM, group=group)
    stats9[:-3] = stats9[:-3] / world_size
    return stats9
