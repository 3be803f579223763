"""-m gpu: crct.ddp.FlatGradDDP with a REAL second rank (train.py:138-143,181-189,205-215).

Two fresh worker processes (tests/ddp_worker.py) share cuda:0 and exchange over gloo; they are started by
tests/conftest.py at session start -- before this pytest process initialises the GPU -- and run beside the other GPU tests;
this module waits for them and checks what rank 0 and rank 1 wrote against the CPU oracle:

  * the all-reduced flat gradient (event mode: one engine backward call, bucketed all-reduces behind per-segment events)
    equals the oracle's gradient of the loss over the CONCATENATED batch (= the mean of the two ranks' losses),
  * ``no_sync()`` + ``batch_multiply = 2`` gives the same gradient, and nothing is exchanged on the first micro-step,
  * both ranks hold identical gradients, rank 1 received rank 0's parameters at construction,
  * the 9-float stats all-reduce returns the world average / sums.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crct import config as C                       # noqa: E402
from crct import synthetic as S                    # noqa: E402
from oracle import crct_oracle as O                # noqa: E402
from helpers import seeded_weights                 # noqa: E402


def _results(case, ddp_workers):
    if ddp_workers is None:
        pytest.skip("the DDP workers were not started (no GPU visible at session start)")
    outdir, procs = ddp_workers
    if case not in procs:
        pytest.skip("needs two GPUs: the RCCL workers (one rank per GPU) are only started where torch.cuda.device_count() >= 2")
    for pr in procs[case]:
        rc = pr.wait(timeout=240)
        assert rc == 0, "DDP worker failed:\n" + open(pr.log_path).read()[-4000:]
    res = []
    for r in range(2):
        with np.load(os.path.join(outdir, "%s_rank%d.npz" % (case, r))) as z:
            res.append({k: z[k] for k in z.files})      # read every array ONCE (an NpzFile re-reads the file on each access)
    return res


def _cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


# "*_rccl": the same worker with one rank per GPU and the product's default route (crct.rccl: ncclCommInitRank over two
# processes, ncclAllReduce / the stats all-reduce on the engine's auxiliary stream) -- only where the box has two GPUs
@pytest.mark.parametrize("case", ["tiny", "full", "tiny_rccl", "full_rccl"])
def test_two_rank_gradient_exchange_matches_oracle(case, ddp_workers):
    r0, r1 = _results(case, ddp_workers)
    case = case.split("_")[0]
    if case == "tiny":
        cfg = C.tiny_config()
        params = C.default_params(categories=9)
        batch = S.make_batch(8, 9, 6, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=5)
    else:
        cfg = C.vilbert_config(v_feature_size=2048, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                               v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
        params = C.default_params()
        batch = S.make_batch(16, 20, 36, 2048, seed=5)
        torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    cpu_params = dict(params, device=torch.device("cpu"))
    sd = seeded_weights(cfg, cpu_params, base_seed=7)           # rank 0's weights: rank 1 must have received them
    ref = O.oracle_step(sd, cfg, cpu_params, batch, cls_dropout=0.0)
    ref[0].backward()
    from crct.layout import parameter_table
    table, total = parameter_table(cfg, cpu_params)
    assert int(r0["n_buckets"][0]) >= (4 if case == "tiny" else 8)
    # train.py:138-143 semantics: the collectives go out WHILE backward is being issued -- every bucket was launched from the
    # engine's per-segment callback, before the backward call returned (round 2 queued them all after it)
    for r in (r0, r1):
        assert int(r["issued_inside_call"][0]) == int(r["n_buckets"][0]) == int(r["issued_inside_call_bf16"][0])
    # identical on both ranks, and equal to the mean of the ranks' losses' gradient
    for key in ("g_sync", "g_accum", "g_bf16", "g_materialized", "params_after_step", "stats9_async"):
        assert np.array_equal(r0[key], r1[key]), key
    # materialize_grads=False: what the fp32 .grad views hold is the SAME on both ranks (reduced values where backward accumulates in
    # fp32, NaN for the Linear weights) -- round 3 left each rank's LOCAL gradient there
    assert np.array_equal(r0["g_bf16_local_fp32"], r1["g_bf16_local_fp32"], equal_nan=True)
    assert np.array_equal(r0["stats9_async"], r0["stats9"])
    mean_loss = 0.5 * (float(r0["loss"][0]) + float(r1["loss"][0]))
    assert abs(mean_loss - float(ref[0])) <= 2e-2 * abs(float(ref[0]))
    assert np.array_equal(r0["params_after_broadcast"], r1["params_after_broadcast"])
    bad = []
    for e in table:
        if not e.used:
            continue
        r = sd[e.name].grad.flatten()
        rn = float(r.double().norm())
        for key in ("g_sync", "g_accum", "g_bf16"):       # g_bf16: each rank's gradient rounded to bf16 before the sum, summed in bf16
            g = torch.from_numpy(r0[key][e.offset:e.offset + e.numel])
            if rn < 1e-7:
                assert float(g.double().norm()) < 1e-3, (e.name, key)
                continue
            c, ratio = _cosine(g, r), float(g.double().norm()) / rn
            # tiny: 0.99 as in test_tiny_step_matches_reference; full depth: the bounds of test_full_size_step_matches_oracle
            tol_c, tol_r = (0.99, 0.08) if case == "tiny" else (0.97, 0.06)
            if c < tol_c or abs(ratio - 1) > tol_r:
                bad.append((key, e.name, round(c, 4), round(ratio, 4)))
    assert not bad, (len(bad), bad[:10])
    # the bf16 payload against the fp32 payload of the same step: 8 significant bits per addend
    used = np.concatenate([np.arange(e.offset, e.offset + e.numel) for e in table if e.used])
    a, b = torch.from_numpy(r0["g_bf16"][used]).double(), torch.from_numpy(r0["g_sync"][used]).double()
    assert float((a - b).norm() / b.norm()) < 8e-3
    # materialize_grads: what the fp32 .grad views hold afterwards is that bf16 result (rewritten for this pass's batch: equal
    # to g_bf16 up to the dropout-free determinism of the step)
    assert np.array_equal(r0["g_materialized"][used], r0["g_bf16"][used])
    # what .grad holds after a DEFAULT bf16 exchange.  No fused optimizer attached: every view holds the reduced gradient
    assert np.array_equal(r0["g_default_grad_views"][used], r0["g_bf16"][used]) and np.array_equal(r1["g_default_grad_views"][used], r0["g_bf16"][used])
    # FusedAdamW attached: the views backward accumulates in fp32 hold the reduced gradient, the owned Linear weights' read NaN
    # (their reduced values live in the bf16 buffer the optimizer reads) -- on both ranks, never a local or a stale value
    for r in (r0, r1):
        v, g16 = r["g_views_with_fused_optimizer"][used], r["g_bf16_with_fused_optimizer"][used]
        nan = np.isnan(v)
        assert 0.80 < nan.mean() < 0.99, nan.mean()                       # the owned Linear weights: ~90 % of the elements (the word table is not one of them)
        assert np.array_equal(v[~nan], g16[~nan])
        assert np.array_equal(g16, r0["g_bf16"][used])                    # same batch, same weights: the same reduced gradient as step (3)
    for e in table:                                                        # NaN exactly on 2-D encoder / head Linear weights, nowhere else
        if e.used and (e.name.endswith(".bias") or "LayerNorm" in e.name or "embeddings" in e.name.split("encoder")[0]):
            if not e.name.endswith("new_image_embeddings.weight"):
                assert not np.isnan(r0["g_views_with_fused_optimizer"][e.offset:e.offset + e.numel]).any(), e.name
    # after a packed (non-direct) pass with materialize_grads=False the owned views are NaN as well, on both ranks alike
    for r in (r0, r1):
        loc = r["g_bf16_local_fp32"][used]
        assert np.isnan(loc).mean() > 0.80 and np.array_equal(loc[~np.isnan(loc)], r["g_bf16"][used][~np.isnan(loc)])
    # weight gradients written into the communication buffer by the GEMMs themselves (after a lazy clear): the same bits
    assert int(r0["packed_runs_only_after_full_clear"][0]) == 0 and int(r0["packed_runs_only_after_lazy_clear"][0]) == 1
    assert np.array_equal(r0["g_bf16_direct"][used], r0["g_bf16"][used]) and np.array_equal(r1["g_bf16_direct"][used], r0["g_bf16_direct"][used])
    # the AdamW step from the bf16 buffer moved the parameters (both ranks identically: asserted above)
    assert not np.array_equal(r0["params_after_step"][:4096], r0["params_after_broadcast"])
    # no exchange on the accumulation-only micro-step: the two ranks' local gradients differ
    assert not np.array_equal(r0["g_local_after_no_sync"], r1["g_local_after_no_sync"])
    # stats: first six averaged over the ranks, last three summed (train.py:181-189)
    s = r0["stats9"]
    assert np.array_equal(s, r1["stats9"])
    assert abs(float(s[0]) - mean_loss) < 1e-5
    needs = batch["R"][:, 1] == 1
    assert float(s[6]) == float(needs.sum())
