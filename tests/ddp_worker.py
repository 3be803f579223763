"""One rank of the 2-rank data-parallel GPU test (tests/test_ddp_gpu.py).  Both ranks share cuda:0 (the GPU box has one GPU) and
exchange over gloo, as bench.py's CRCT_BENCH_SHARE_GPU path does; the exchange code under test -- crct.ddp.FlatGradDDP in event
mode: ONE engine backward call, per-segment events, bucketed all-reduces on a communication stream -- is the one a real
multi-GPU run uses with the RCCL backend.  Each rank takes its half of a batch; rank 0 writes the all-reduced gradients.

    python tests/ddp_worker.py <rank> <world> <port> <outdir> <case> [backend]

backend "nccl" (a box with >= 2 GPUs: tests/test_ddp_gpu.py starts it only there): rank r runs on cuda:r and the exchange is the
product's default route -- crct.rccl.Communicator over the ranks, ncclAllReduce on the engine's auxiliary stream.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def halves(batch, rank, world):
    B = batch["tokens"].shape[0]
    per = B // world
    return {k: (v[rank * per:(rank + 1) * per].clone() if hasattr(v, "shape") and v.shape[0] == B else v) for k, v in batch.items()}


def main():
    rank, world, port, outdir, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from crct import config as C, synthetic as S
    from crct.model import VisualDialogEncoder
    from crct.step_adapter import forward as step_forward
    from crct.ddp import FlatGradDDP, all_reduce_stats

    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if case == "tiny":
        cfg = C.tiny_config()
        params = C.default_params(categories=9, device=dev, rank=rank, world_size=world, ddp=True)
        batch = S.make_batch(8, 9, 6, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=5)
        bucket_mb = 0.05
    else:
        cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                               v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
        params = C.default_params(device=dev, rank=rank, world_size=world, ddp=True)
        batch = S.make_batch(16, 20, 36, 2048, seed=5)
        bucket_mb = 64
    model = VisualDialogEncoder(params, config=cfg)
    core = model.bert_pretrained
    core.cls_dropout = 0.0
    # rank 1 starts from DIFFERENT weights: the constructor's broadcast from rank 0 (train.py:139 semantics) must fix that
    S.seeded_fill_(model.state_dict(), base_seed=7 if rank == 0 else 8)
    core._invalidate_shadow()
    # the reference's payload first (fp32: the numbers of steps (1) and (2) are compared with the oracle at the fp32 bounds)
    ddp = FlatGradDDP(model, bucket_mb=bucket_mb, grad_dtype=torch.float32)
    assert ddp.event_mode
    assert (ddp.communicator() is not None) == (backend == "nccl")      # RCCL called directly on the auxiliary stream, or torch.distributed (gloo)
    mine = halves(batch, rank, world)
    out = {}

    # (1) one synchronised step: every rank ends up with the gradient of the mean loss over the whole batch
    core.zero_flat_grads()
    res = step_forward(model, mine, params)
    res[0].backward()
    torch.cuda.synchronize()
    out["g_sync"] = core.flat_grads.detach().cpu().numpy().copy()
    out["loss"] = np.array([float(res[0])])
    out["n_buckets"] = np.array([len(ddp._buckets)])
    # every bucket's collective was launched from the engine's per-segment callback, i.e. before the backward call returned
    out["issued_inside_call"] = np.array([ddp.issued_inside_engine_call])
    st = core.last_stats[8:17].clone()
    all_reduce_stats(st, world)
    out["stats9"] = st.cpu().numpy()

    # (2) batch_multiply = 2 (train.py:205-215): two micro-steps on quarter batches, exchange only on the second
    quarters = [halves(mine, i, 2) for i in range(2)]
    core.zero_flat_grads()
    with ddp.no_sync():
        (step_forward(model, quarters[0], params)[0] / 2).backward()
    torch.cuda.synchronize()
    out["g_local_after_no_sync"] = core.flat_grads.detach().cpu().numpy().copy()
    (step_forward(model, quarters[1], params)[0] / 2).backward()
    torch.cuda.synchronize()
    out["g_accum"] = core.flat_grads.detach().cpu().numpy().copy()
    out["params_after_broadcast"] = core.flat_params.detach().cpu().numpy().copy()[:4096]

    # (3) the bf16 payload (default).  No fused optimizer is attached yet, so by default the reduced bucket is written back into
    # the fp32 .grad views (what a stock optimizer / clip_grad_norm_ / a GradScaler read: ADVICE r3) ...
    from crct.optim import get_optimizer
    from crct.ddp import AsyncStats
    ddp16 = FlatGradDDP(model, bucket_mb=bucket_mb, broadcast=False)
    assert ddp16.grad_dtype == torch.bfloat16 and core._ddp is ddp16 and ddp16.materializes()
    core.zero_flat_grads()
    step_forward(model, mine, params)[0].backward()
    assert ddp16.grad_source() is None
    torch.cuda.synchronize()
    out["g_default_grad_views"] = core.flat_grads.detach().cpu().numpy().copy()
    # ... and with materialize_grads=False the all-reduced gradient lives in the communication buffer only
    ddp16.materialize_grads = False
    core.zero_flat_grads()
    red = AsyncStats(world, device=dev)
    res = step_forward(model, mine, params)
    red.launch(core.last_stats[8:17])
    res[0].backward()
    src = ddp16.grad_source()
    assert src is not None and src.dtype == torch.bfloat16
    torch.cuda.synchronize()
    out["g_bf16"] = src.float().cpu().numpy().copy()
    out["g_bf16_local_fp32"] = core.flat_grads.detach().cpu().numpy().copy()
    out["issued_inside_call_bf16"] = np.array([ddp16.issued_inside_engine_call])
    out["stats9_async"] = red.result().cpu().numpy()
    out["packed_runs_only_after_full_clear"] = np.array([int(ddp16.packed_runs_only)])
    # (3b) after a LAZY clear the pass writes the owned weight gradients: the engine puts them into the communication buffer itself
    # (bf16 from the GEMM epilogue) and the exchange packs only the accumulated runs -- the same bits as packing everything
    core.zero_flat_grads(lazy=True)
    step_forward(model, mine, params)[0].backward()
    torch.cuda.synchronize()
    out["g_bf16_direct"] = ddp16.grad_source().float().cpu().numpy().copy()
    out["packed_runs_only_after_lazy_clear"] = np.array([int(ddp16.packed_runs_only)])
    # (4) materialize_grads: the all-reduced bf16 bucket is written back into the fp32 .grad views (same weights, same batch)
    ddp16.materialize_grads = True
    core.zero_flat_grads()
    step_forward(model, mine, params)[0].backward()
    assert ddp16.grad_source() is None
    torch.cuda.synchronize()
    out["g_materialized"] = core.flat_grads.detach().cpu().numpy().copy()
    # (5) the fused AdamW consumes the bf16 buffer as it lies: one step, both ranks must end on identical parameters.  With it
    # attached the default (materialize_grads=None) no longer writes the weight gradients back: .grad of what backward accumulates
    # in fp32 holds the reduced values, the fp32 views of the Linear weight gradients read NaN -- never a local or stale gradient
    ddp16.materialize_grads = None
    opt = get_optimizer(params, model)
    assert not ddp16.materializes()
    core.zero_flat_grads(lazy=True)
    step_forward(model, mine, params)[0].backward()
    assert ddp16.grad_source() is not None
    torch.cuda.synchronize()
    out["g_views_with_fused_optimizer"] = core.flat_grads.detach().cpu().numpy().copy()
    out["g_bf16_with_fused_optimizer"] = ddp16.grad_source().float().cpu().numpy().copy()
    opt.overlap = case != "tiny"
    opt.step()
    opt.zero_grad()
    opt.synchronize()
    torch.cuda.synchronize()
    out["params_after_step"] = core.flat_params.detach().cpu().numpy().copy()
    np.savez(os.path.join(outdir, "%s_rank%d.npz" % (case, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
