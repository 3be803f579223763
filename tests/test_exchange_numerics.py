"""What a bf16 SUM all-reduce over 8 ranks does to the CRCT gradients (CPU emulation).

The data-parallel exchange (crct/ddp.py) ships bf16: every rank rounds its local gradient (already scaled by 1 / world) to bf16,
RCCL sums in bf16 -- a ring reduce-scatter adds the local shard to the running partial at each of world - 1 hops and the partial
travels as bf16, so a chunk is rounded world - 1 times on its way, in an order that differs per chunk -- and AdamW reads the
bf16 sums.  The 2-rank GPU test (tests/test_ddp_gpu.py) bounds one rounding of the sum; no 8-GPU node has run this code.  This
test states the bound for world = 8 on REAL gradient statistics: eight different batches through the fp32 oracle at the tiny
configuration give eight per-rank gradients; the ring (and a binary tree, the other order RCCL uses) is emulated with a rounding
per hop and compared with the fp32 mean of the fp32 gradients, per parameter tensor.

Measured here (worst parameter tensor): ring cosine 0.999987, relative L2 error 5.2e-3; tree 0.999991 / 4.6e-3; for scale, ONE
rounding of the exact sum to bf16 is 0.999998 / 3.1e-3, and the bf16 STEP itself agrees with the fp32 oracle to cosine 0.983 - 0.987
(tests/test_step_gpu.py): the 8-rank exchange adds 1.7 x the error of a single rounding, three orders of magnitude below the
arithmetic's 1 - cosine.  (An fp32 payload, `grad_dtype=torch.float32`, has none of this and costs 0.1 ms more on one GPU.)
"""
import torch

from crct import config as CFG
from crct import synthetic as S
from helpers import seeded_weights
from oracle import crct_oracle as O

WORLD = 8


def _rank_gradients():
    cfg, params = CFG.tiny_config(), CFG.default_params(categories=9)
    params["device"] = torch.device("cpu")
    grads = []
    for r in range(WORLD):
        sd = seeded_weights(cfg, params)
        batch = S.make_batch(4, 8, 6, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=4321 + 97 * r)
        loss = O.oracle_step(sd, cfg, params, batch)[0]
        loss.backward()
        grads.append({k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone() for k, p in sd.items()})
    return grads


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _ring_sum(payloads, n_chunks=WORLD):
    """Ring reduce-scatter + all-gather of `payloads` (one flat bf16-valued fp32 tensor per rank): chunk c starts at rank c + 1,
    visits the ranks in ring order, every hop computes fp32(partial) + fp32(local) and rounds the result to bf16."""
    n = payloads[0].numel()
    out = torch.empty(n)
    bounds = [n * c // n_chunks for c in range(n_chunks + 1)]
    for c in range(n_chunks):
        lo, hi = bounds[c], bounds[c + 1]
        start = (c + 1) % WORLD
        part = payloads[start][lo:hi].clone()
        for hop in range(1, WORLD):
            part = _bf16(part + payloads[(start + hop) % WORLD][lo:hi])
        out[lo:hi] = part
    return out


def _tree_sum(payloads):
    level = list(payloads)
    while len(level) > 1:
        level = [_bf16(level[i] + level[i + 1]) for i in range(0, len(level), 2)]
    return level[0]


def test_bf16_sum_over_eight_ranks_stays_far_inside_the_step_tolerance():
    grads = _rank_gradients()
    names = sorted(grads[0])
    sizes = [grads[0][k].numel() for k in names]
    flat = [torch.cat([g[k].reshape(-1) for k in names]) / WORLD for g in grads]       # local gradients scaled by 1 / world (ddp.py)
    exact = torch.stack(flat).to(torch.float64).sum(0).to(torch.float32)               # what an fp32 exchange would deliver
    payloads = [_bf16(f) for f in flat]                                                # the pack kernel / the GEMM epilogue round once
    results = {"ring": _ring_sum(payloads), "tree": _tree_sum(payloads)}
    worst = {}
    for name, got in results.items():
        cos_min, rel_max, off = 1.0, 0.0, 0
        for k, n in zip(names, sizes):
            a, b = exact[off:off + n], got[off:off + n]
            off += n
            na = float(a.norm())
            if na == 0.0:                                                              # unused parameters: zero in, zero out
                assert float(b.abs().max()) == 0.0, k
                continue
            cos_min = min(cos_min, float(torch.dot(a, b) / (na * float(b.norm()))))
            rel_max = max(rel_max, float((a - b).norm()) / na)
        worst[name] = (cos_min, rel_max)
    # bf16 keeps 8 bits of mantissa: one rounding is <= 2^-9 relative per element, ~2.3e-3 RMS over a tensor; the payload rounding
    # plus 7 hops of a random-sign walk stays below 1e-2 of the tensor's norm
    for name, (cos_min, rel_max) in worst.items():
        assert cos_min >= 0.9999, (name, cos_min, rel_max)
        assert rel_max <= 1.0e-2, (name, cos_min, rel_max)
    # the exchange must not bias the sum: the mean signed error over all elements is far below the rounding's own size
    for name, got in results.items():
        err = (got - exact)
        assert abs(float(err.sum())) <= 1e-3 * float(exact.abs().sum()), name
