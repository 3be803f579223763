"""One rank of the 2-rank data-parallel GPU test (tests/test_ddp_gpu.py).  Both ranks share cuda:0 (the GPU box has one GPU) and
exchange over gloo, as bench.py's CRCT_BENCH_SHARE_GPU path does; the exchange code under test -- crct.ddp.FlatGradDDP in event
mode: ONE engine backward call, per-segment events, bucketed all-reduces on a communication stream -- is the one a real
multi-GPU run uses with the RCCL backend.  Each rank takes its half of a batch; rank 0 writes the all-reduced gradients.

    python tests/ddp_worker.py <rank> <world> <port> <outdir> <case>
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
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from crct import config as C, synthetic as S
    from crct.model import VisualDialogEncoder
    from crct.step_adapter import forward as step_forward
    from crct.ddp import FlatGradDDP, all_reduce_stats

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
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
    ddp = FlatGradDDP(model, bucket_mb=bucket_mb)
    assert ddp.event_mode
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
    np.savez(os.path.join(outdir, "%s_rank%d.npz" % (case, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
