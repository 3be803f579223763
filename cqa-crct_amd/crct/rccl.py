"""RCCL called directly (ctypes on the librccl.so torch itself ships and has already loaded), for ONE reason: the collective
must run on a HIP stream this package chose.

``torch.distributed``'s NCCL process group enqueues every collective on a stream of its own.  MI355X schedules all HIP streams
of a process onto 4 hardware queues, and streams that share a queue are serialised (``csrc/streams.hip``); where the process
group's stream lands is an accident of creation order.  Measured on one MI355X with a single-rank communicator, i.e. with
collectives that move nothing: 10 ``dist.all_reduce`` calls per step cost 0.66 ms of GPU time, because the group's stream sat
on the queue of one of the engine's compute streams and its event waits stalled that queue (profiles/r3_ddp_stream_placement.txt);
with real transfers the collectives themselves would run inside a compute stream's queue.  Called directly, ``ncclAllReduce``
takes the stream as an argument: the engine's auxiliary stream (``crct_engine_aux_stream``), which has a hardware queue to itself.

Bootstrap: rank 0 draws the ``ncclUniqueId`` and the existing ``torch.distributed`` group (any backend) broadcasts its 128
bytes -- the only thing torch.distributed is used for on this path (plus what the training loop itself does with it).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
ncclSum = 0
DTYPES = {torch.float32: 7, torch.bfloat16: 9, torch.float16: 6, torch.int32: 2, torch.int64: 4, torch.uint8: 1}      # rccl.h ncclDataType_t


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            raise RuntimeError("librccl.so not found next to torch (%s): the data-parallel exchange needs RCCL" % path)
        lib = C.CDLL(path)
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclGetVersion.argtypes = [C.POINTER(C.c_int)]
        lib.ncclCommGetAsyncError.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib = lib
    return _lib


def version():
    """RCCL's own version code (ncclGetVersion: major * 10000 + minor * 100 + patch for 2.x >= 2.9) as (code, "x.y.z")."""
    v = C.c_int(0)
    _check(_load().ncclGetVersion(C.byref(v)), "ncclGetVersion")
    code = v.value
    return code, "%d.%d.%d" % (code // 10000, (code // 100) % 100, code % 100)


def env_seen():
    """The NCCL_* / RCCL_* / HSA_* variables this process runs the collectives under (reported by bench.py next to every
    multi-GPU number: channel counts, protocols and algorithms are chosen by RCCL from these and the topology)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HSA_FORCE_FINE_GRAIN_PCIE")}


def channels_from_debug_log(path):
    """Channel count of the communicator as RCCL itself logged it (NCCL_DEBUG=INFO with NCCL_DEBUG_FILE=path set before the
    communicator was created): the public API has no getter.  None when the log has no such line."""
    import re
    try:
        with open(path) as f:
            text = f.read()
    except OSError:
        return None
    best = None
    for m in re.finditer(r"(\d+) coll channels", text):
        best = max(best or 0, int(m.group(1)))
    if best is None:
        for m in re.finditer(r"Channel (\d+)/(\d+)", text):
            best = max(best or 0, int(m.group(2)))
    return best


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("RCCL %s failed: %s" % (what, _load().ncclGetErrorString(rc).decode()))


class Communicator(object):
    """One RCCL communicator over the ranks of ``group`` (default: the world), created on ``device``."""

    def __init__(self, device, group=None):
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device)
        # Every stage that can fail on ONE rank is agreed on by ALL ranks before anyone enters the next collective: a rank that
        # raised here while its peers went on into the broadcast / ncclCommInitRank below would leave them waiting for ever.
        # Stage 1: the library loads everywhere (MIN over an ok flag).  Stage 2: rank 0's ncclGetUniqueId result travels in the
        # broadcast itself (the id, or the error text).  Stage 3, ncclCommInitRank, is itself the rendezvous: a failure of only
        # SOME ranks inside it cannot be recovered from here (the others block in RCCL's bootstrap until its own timeout).
        lib, load_err = None, None
        try:
            lib = _load()
        except Exception as e:              # noqa: BLE001 -- reported to every rank below
            load_err = e
        if self.world > 1:
            flag_dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            ok = torch.tensor([0 if lib is None else 1], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) != 1:
                raise RuntimeError("RCCL could not be loaded on %s: %r" % ("this rank" if lib is None else "a peer rank", load_err))
        elif lib is None:
            raise load_err
        uid = _UniqueId()
        box = [None]
        if self.rank == 0:
            rc = lib.ncclGetUniqueId(C.byref(uid))
            # a failure here is BROADCAST (as a string) instead of raised at once: the peers are about to wait in the broadcast below
            box = [C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES) if rc == 0      # the raw 128 bytes (.internal would stop at a NUL)
                   else "ncclGetUniqueId failed on rank 0: %s" % lib.ncclGetErrorString(rc).decode()]
        if self.world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
        raw = box[0]
        if not isinstance(raw, bytes):
            raise RuntimeError("RCCL %s" % raw)
        if len(raw) != NCCL_UNIQUE_ID_BYTES:       # c_char arrays stop at the first NUL when read as .value: take the raw buffer
            raise RuntimeError("ncclUniqueId has %d bytes" % len(raw))
        C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        n = C.c_int(-1)
        _check(lib.ncclCommCount(self.comm, C.byref(n)), "ncclCommCount")
        if n.value != self.world:
            self.destroy()                  # not leaked: the caller falls back to torch.distributed's group
            raise RuntimeError("RCCL communicator spans %d ranks, expected %d" % (n.value, self.world))
        self.collectives = 0

    def check_async(self):
        """A collective's enqueue can succeed and the operation still fail later (a peer died, a transport error): RCCL
        reports that through ncclCommGetAsyncError.  Polled by the exchange at the end of every backward pass (ddp.py) --
        a host call on the communicator's state, no GPU synchronisation."""
        err = C.c_int(0)
        _check(_load().ncclCommGetAsyncError(self.comm, C.byref(err)), "ncclCommGetAsyncError")
        if err.value != 0:
            raise RuntimeError("RCCL reported an asynchronous error on rank %d after %d collectives: %s"
                               % (self.rank, self.collectives, _load().ncclGetErrorString(err.value).decode()))

    def all_reduce_(self, t, stream):
        """In-place SUM of a contiguous CUDA tensor, enqueued on ``stream`` (a torch stream); returns at once."""
        if not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError("all_reduce_: contiguous CUDA tensor required")
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), DTYPES[t.dtype], ncclSum, self.comm, stream.cuda_stream), "ncclAllReduce")
        self.collectives += 1

    def broadcast_(self, t, root, stream):
        if not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError("broadcast_: contiguous CUDA tensor required")
        _check(_load().ncclBroadcast(t.data_ptr(), t.data_ptr(), t.numel(), DTYPES[t.dtype], int(root), self.comm, stream.cuda_stream), "ncclBroadcast")
        self.collectives += 1

    def destroy(self):
        """Explicit only (not from __del__: at interpreter exit the HIP runtime may already be gone)."""
        if getattr(self, "comm", None):
            _load().ncclCommDestroy(self.comm)
            self.comm = None
