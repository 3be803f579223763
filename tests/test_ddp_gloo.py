"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (crct/ddp.py) -- bucket planning over
the engine's backward segments, all-reduce launched as segments finish, 1/world averaging, and the
reference's 9-float stats all-reduce (train.py:181-189).  The GPU path only swaps the segment runner
(native backward) and the backend (RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from crct import config as CFG
from crct import layout as LY
from crct.ddp import plan_buckets, reduce_while_running, all_reduce_stats, BucketExchange, AsyncStats


def _segments(cfg, params):
    """Backward-ordered gradient ranges, as crct_engine_segment_range reports them."""
    table, _ = LY.parameter_table(cfg, params)

    def rng(prefixes):
        es = [e for e in table if e.used and any(e.name.startswith(p) for p in prefixes)]
        return (min(e.offset for e in es), max(e.offset + e.numel for e in es))
    segs = [rng(["bert.t_pooler.", "bert.v_pooler.", "cls.bi_seq_relationship.", "regressor."])]
    for kind, i in reversed(LY.encoder_schedule(cfg)):
        segs.append(rng([{"t": "bert.encoder.layer.%d.", "v": "bert.encoder.v_layer.%d.", "c": "bert.encoder.c_layer.%d."}[kind] % i]))
    segs.append(rng(["bert.embeddings.", "bert.v_embeddings."]))
    return segs, table


def test_bucket_plan_covers_every_used_gradient_once():
    cfg, params = CFG.vilbert_config(), CFG.default_params()
    segs, table = _segments(cfg, params)
    assert len(segs) == 26
    # backward order = descending offsets, contiguous ranges that do not overlap
    for (lo0, hi0), (lo1, hi1) in zip(segs, segs[1:]):
        assert hi1 <= lo0
    buckets = plan_buckets(segs, 64 * (1 << 20) // 4)
    assert 8 <= len(buckets) <= 20
    covered = torch.zeros(max(hi for _, hi in segs), dtype=torch.int8)
    for _, lo, hi in buckets:
        covered[lo:hi] += 1
    assert int(covered.max()) == 1
    for e in table:
        c = covered[e.offset:e.offset + e.numel]
        if e.used:
            assert c.numel() == e.numel and bool((c == 1).all()), e.name
        else:
            assert not bool((c == 1).any()), e.name
    # a bucket becomes ready exactly when its last segment has run
    assert [b[0] for b in buckets] == sorted(b[0] for b in buckets) and buckets[-1][0] == len(segs) - 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, params = CFG.tiny_config(), CFG.default_params(categories=9)
    segs, table = _segments(cfg, params)
    total = max(hi for _, hi in segs)
    flat = torch.zeros(total)
    order = []

    def run_segment(i):                       # stand-in for crct_engine_backward(seg=i): local grads scaled by 1/world
        lo, hi = segs[i]
        flat[lo:hi] += (rank + 1.0) * (i + 1) / world
        order.append(i)

    buckets = plan_buckets(segs, 200000)
    reduce_while_running(flat, segs, buckets, run_segment)
    ok = order == list(range(len(segs)))
    for i, (lo, hi) in enumerate(segs):       # SUM of (r+1)(i+1)/world over ranks = (i+1) * mean(r+1)
        ok = ok and bool(torch.allclose(flat[lo:hi], torch.full((hi - lo,), (i + 1) * (world + 1) / 2.0)))
    stats = torch.tensor([1.0 + rank, 2, 3, 4, 5, 6, 7 + rank, 8, 9])
    all_reduce_stats(stats, world)
    ok = ok and bool(torch.allclose(stats, torch.tensor([1.5, 2, 3, 4, 5, 6, 15.0, 16, 18])))

    # ---- the exchange as FlatGradDDP drives it: buckets launched from the per-segment callback WHILE "backward" is still
    # running, bf16 payload in a communication buffer with the flat buffer's offsets, optional write-back into the fp32 buffer
    for materialize in (False, True):
        flat2 = torch.zeros(total)
        comm = torch.zeros(total, dtype=torch.bfloat16)
        launched_at = []
        ex = BucketExchange(flat2, buckets, None, comm_buf=comm, materialize=materialize,
                            pack=lambda src, dst: dst.copy_(src.to(torch.bfloat16)), unpack=lambda src, dst: dst.copy_(src.float()))
        last_of = {last: b for b, (last, _, _) in enumerate(buckets)}
        for i in range(len(segs)):                 # the engine: enqueue segment i, then call back
            lo, hi = segs[i]
            flat2[lo:hi] += (rank + 1.0) * (i + 1) / world
            if i in last_of:
                ex.launch(last_of[i])
                launched_at.append(i)
        n_inside = len(ex.issue_order)
        ex.finish()
        ok = ok and n_inside == len(buckets) and ex.issue_order == list(range(len(buckets))) and launched_at == [b[0] for b in buckets]
        for i, (lo, hi) in enumerate(segs):
            want = torch.full((hi - lo,), (i + 1) * (world + 1) / 2.0)
            ok = ok and bool(torch.allclose(comm[lo:hi].float(), want, rtol=1e-2))          # bf16 payload: 8 significant bits
            if materialize:
                ok = ok and bool(torch.equal(flat2[lo:hi], comm[lo:hi].float()))
            else:                                                                           # the fp32 buffer keeps the LOCAL gradient
                ok = ok and bool(torch.allclose(flat2[lo:hi], torch.full((hi - lo,), (rank + 1.0) * (i + 1) / world)))
    # ---- asynchronous stats all-reduce (train.py:181-189 off the critical path)
    red = AsyncStats(world)
    src = torch.tensor([1.0 + rank, 2, 3, 4, 5, 6, 7 + rank, 8, 9])
    red.launch(src)
    src.zero_()                                     # the caller's tensor is free again at once: the reducer works on its copy
    ok = ok and bool(torch.allclose(red.result(), torch.tensor([1.5, 2, 3, 4, 5, 6, 15.0, 16, 18]))) and red.result() is None
    q.put((rank, ok, len(buckets)))
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] >= 2


def test_what_grad_holds_follows_the_optimizer_that_is_alive():
    """`materialize_grads=None` writes the reduced gradients back into the fp32 views UNLESS a live FusedAdamW that covers
    every gradient is attached -- an optimizer that was built and discarded, or one over a parameter subset, must not switch the
    write-back off; and a pass without an exchange (no_sync / one rank) forgets that the owned views were NaN-filled."""
    import weakref
    from crct.ddp import FlatGradDDP

    class Core(object):
        pass

    class Opt(object):
        def __init__(self, full):
            self.full = full

        def covers_every_gradient(self):
            return self.full
    ddp = object.__new__(FlatGradDDP)
    ddp.core, ddp.materialize_grads = Core(), None
    assert ddp.materializes() is True                       # nothing attached: DistributedDataParallel's contract
    full = Opt(True)
    ddp.core._fused_optimizer = weakref.ref(full)
    assert ddp.materializes() is False                      # the fused optimizer reads the bf16 bucket itself
    part = Opt(False)
    ddp.core._fused_optimizer = weakref.ref(part)
    assert ddp.materializes() is True                       # a subset optimizer: somebody else reads .grad of the rest
    ddp.core._fused_optimizer = weakref.ref(full)
    del full
    import gc
    gc.collect()
    assert ddp.materializes() is True                       # built, then discarded
    ddp.materialize_grads = False
    assert ddp.materializes() is False                      # an explicit choice stands
