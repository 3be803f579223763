"""Data-parallel gradient exchange for the CRCT step: RCCL all-reduce of the flat gradient buffer,
overlapped with backward.

Reference behaviour being replaced: ``DistributedDataParallel(model, device_ids=[gpu],
find_unused_parameters=True)`` (CRCT/train.py:138-143) = parameter broadcast from rank 0 at
construction + bucketed gradient all-reduce (AVG) triggered by autograd hooks, with the 36
never-used tensors tolerated.  Here:

  * parameters live in one flat buffer ordered by first use, so each backward segment of the
    native engine (heads, then the encoder schedule reversed, then the embeddings) completes one
    CONTIGUOUS range of the flat gradient buffer;
  * consecutive segments are merged into buckets of >= ``bucket_mb``.  Backward is ONE engine call;
    the engine records events on its internal streams after every segment, and the ``all_reduce(SUM)``
    of a bucket is queued on a communication stream behind the events of the bucket's last segment
    -- the RCCL process group runs it on its own HIP stream, so the exchange of bucket k overlaps the
    backward kernels of bucket k+1 (``event_mode=False`` falls back to one engine call per segment
    with the collectives launched in between; measured 21 % slower per step on one MI355X);
  * the 1/world averaging is folded into the loss-gradient seeds (no extra pass over 953 MB);
  * tensors that never receive a gradient sit at the tail of the layout and are never sent.

xGMI note (MI355X: 7 links x ~153 GB/s per GPU, point-to-point): few large messages let RCCL use all
links; the default 64 MB buckets give ~15 collectives per step.
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
    have been ordered before further work on the current stream (no host block on GPU)."""
    works, b = [], 0
    for i in range(len(segments)):
        run_segment(i)
        while b < len(buckets) and buckets[b][0] == i:
            _, lo, hi = buckets[b]
            works.append(dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
            b += 1
    for w in works:
        w.wait()


def reduce_behind_events(flat_grads, buckets, seg_events, comm_stream, group=None):
    """The whole backward has been enqueued in ONE engine call that recorded ``seg_events[4*i .. 4*i+3]`` after
    segment i.  Launch every bucket's all-reduce on ``comm_stream`` behind the events of its last segment; the
    exchange overlaps the backward kernels still running.  The current stream is ordered after all collectives."""
    works = []
    with torch.cuda.stream(comm_stream):
        for last, lo, hi in buckets:
            for ev in seg_events[4 * last:4 * last + 4]:
                comm_stream.wait_event(ev)
            works.append(dist.all_reduce(flat_grads[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()


class FlatGradDDP(object):
    """Attach to a ``VisualDialogEncoder`` / ``CrctModel``: ``FlatGradDDP(model)`` after
    ``dist.init_process_group(backend='nccl', ...)`` (RCCL on ROCm)."""

    def __init__(self, model, process_group=None, bucket_mb=64, broadcast=True):
        from .optim import _crct_core
        self.core = _crct_core(model)
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._buckets = None
        self._events = self._comm = None
        self.event_mode = True       # False: segment-by-segment engine calls with the collectives launched in between
        self.require_sync = True
        self.force_exchange = False  # developer switch: run the bucketed exchange even on a single rank
        if broadcast:      # DDP constructor semantics: rank 0's parameters win (train.py:139)
            dist.broadcast(self.core.flat_params, 0, group=process_group)
            self.core._invalidate_shadow()
        self.core._ddp = self

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (``batch_multiply`` > 1): skip the exchange on all but the last micro-step."""
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    def backward(self, core, eng, tensors, step):
        step = dict(step)
        if self.world > 1:      # also on accumulation-only micro-steps: the final SUM then yields the average
            step["grad_scale"] = step.get("grad_scale", 1.0) / self.world      # folded into the head kernel's gradient seeds
        if not self.require_sync or (self.world == 1 and not self.force_exchange):
            eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, -1)
            return
        if self._buckets is None:
            self._buckets = plan_buckets(eng.segments, self.bucket_elems)
        if self.event_mode and core.flat_grads.is_cuda:
            # one engine call (full overlap of its internal streams); the engine marks the end of every segment with
            # events and the collectives queue up behind them on a communication stream
            if self._events is None:
                self._comm = torch.cuda.Stream(device=core.flat_grads.device)
                # stock torch events (system-scope release at every record): what follows them is a collective whose peers
                # read and write across GPUs -- unlike the engine-internal and optimizer events (crct/events.py), which order
                # streams of one device only
                self._events = [torch.cuda.Event() for _ in range(4 * len(eng.segments))]
                for ev in self._events:          # torch creates the hipEvent lazily, at the first record
                    ev.record()
            step["seg_done_events"] = self._events
            eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, -1)
            reduce_behind_events(core.flat_grads, self._buckets, self._events, self._comm, self.group)
            return
        reduce_while_running(core.flat_grads, eng.segments, self._buckets,
                             lambda i: eng.backward(core.flat_params, core.flat_shadow, core.flat_grads, tensors, step, i),
                             self.group)


def all_reduce_stats(stats9, world_size, group=None):
    """train.py:181-189: SUM a 9-float stats tensor, first six entries averaged."""
    dist.all_reduce(stats9, op=dist.ReduceOp.SUM, group=group)
    stats9[:-3] = stats9[:-3] / world_size
    return stats9
