"""Device-scope ordering events (include/crct_hip.h: crct_event_*) with the small part of the ``torch.cuda.Event`` surface
the step glue uses.

A stock ``torch.cuda.Event`` is a HIP event with the default system-scope fence: every ``record`` makes device memory visible
to the host and to peer devices (cache write-back / invalidation around it).  The optimizer-overlap and bucket events only
order streams of ONE device, so they are created with ``hipEventDisableSystemFence``; measured on MI355X
(tools/handoff_lab.cpp): a cross-stream hand-off 7-8.5 us instead of 10-11 us, and 0.13 ms per training step for the engine's
internal events alone.  Not for anything the host or another GPU reads (the input pipeline's copy events stay torch events).
"""
import torch

from . import lib as L


class DeviceEvent(object):
    def __init__(self):
        self._lib = L.load()
        self.handle = self._lib.crct_event_create()
        if not self.handle:
            raise RuntimeError("crct_event_create failed: %s" % self._lib.crct_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.crct_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @property
    def cuda_event(self):
        """The raw hipEvent_t (what ``torch.cuda.Event.cuda_event`` returns): CrctStepCfg's event arrays take these."""
        return self.handle

    def record(self, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        L.check(self._lib.crct_event_record(self.handle, s.cuda_stream), "event_record")

    def wait(self, stream=None):
        s = torch.cuda.current_stream() if stream is None else stream
        L.check(self._lib.crct_stream_wait_event(s.cuda_stream, self.handle), "stream_wait_event")

    def query(self):
        r = self._lib.crct_event_query(self.handle)
        if r < 0:
            raise RuntimeError("crct_event_query failed: %s" % self._lib.crct_last_error().decode())
        return bool(r)

    def synchronize(self):
        L.check(self._lib.crct_event_synchronize(self.handle), "event_synchronize")


_pool = []
_next = [0]


def order_streams(src, dst, system=False):
    """Everything enqueued on ``src`` so far happens before whatever is enqueued on ``dst`` from now on (``dst.wait_stream(src)``
    without the system-scope fence).  Events come from a small ring: an event is re-recorded only long after its waiters ran.
    ``system=True``: the stock torch ordering (system-scope release / acquire) -- for consumers of data that a PEER device
    has written into local memory (gradients after an all-reduce)."""
    if src.cuda_stream == dst.cuda_stream:
        return
    if system:
        dst.wait_stream(src)
        return
    if len(_pool) < 64:
        _pool.append(DeviceEvent())
        ev = _pool[-1]
    else:
        ev = _pool[_next[0] % 64]
        _next[0] += 1
    ev.record(src)
    ev.wait(dst)
