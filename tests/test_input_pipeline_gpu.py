"""Host -> HBM input pipeline (reference: pageable DataLoader output + synchronous .to(device),
train.py:58-73, encoder_decorator.py:81-116)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from crct import config as C                       # noqa: E402
from crct import synthetic as S                    # noqa: E402
from crct.input_pipeline import DevicePrefetcher   # noqa: E402
from crct.step_adapter import forward as step_forward   # noqa: E402
from helpers import load_case                      # noqa: E402
from test_step_gpu import build_model              # noqa: E402

DEV = "cuda"


def _batches(n, grow=False):
    out = []
    for i in range(n):
        B = 3 + (i % 3 if grow else 0)
        b = S.make_batch(B, 7, 5, 32, categories=9, vocab_size=128, seed=100 + i)
        b["tag"] = "batch-%d" % i                                   # non-tensor entries pass through
        out.append(b)
    return out


@pytest.mark.parametrize("depth", [2, 3])
def test_batches_arrive_intact_and_slots_are_not_overwritten_early(depth):
    """The consumer is deliberately slow (a long spin kernel before it reads its batch): if a slot were refilled
    before its reader ran, the checksums of later batches would leak into earlier ones."""
    host = _batches(9, grow=True)
    sums = []
    for i, b in enumerate(DevicePrefetcher(host, DEV, depth=depth)):
        assert b["tag"] == host[i]["tag"]
        for k, v in host[i].items():
            if torch.is_tensor(v):
                assert b[k].is_cuda and b[k].dtype == v.dtype and tuple(b[k].shape) == tuple(v.shape)
        torch.cuda._sleep(20_000_000)                                # ~10 ms of GPU time before the batch is read
        sums.append(torch.stack([b[k].double().sum() for k in sorted(b) if torch.is_tensor(b[k])]))
    torch.cuda.synchronize()
    for i, s in enumerate(sums):
        want = torch.stack([host[i][k].double().sum() for k in sorted(host[i]) if torch.is_tensor(host[i][k])])
        assert torch.allclose(s.cpu(), want, rtol=0, atol=1e-6), i


def test_one_copy_per_batch_and_buffer_reuse():
    host = _batches(6)
    pf = DevicePrefetcher(host, DEV, depth=2)
    ptrs = []
    for b in pf:
        ptrs.append(b["image_feat"].data_ptr())
    assert len(set(ptrs)) == 2 and ptrs[0] == ptrs[2] == ptrs[4] and ptrs[1] == ptrs[3]
    per_batch = sum(v.numel() * v.element_size() for v in host[0].values() if torch.is_tensor(v))
    assert per_batch <= pf.bytes_copied / 6 <= per_batch + 256 * len(host[0])      # packed, 256-B aligned fields


def test_training_from_host_batches_equals_resident_batches():
    z, meta, cfg, params, batch = load_case("tiny_L1")
    host = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()} for _ in range(4)]
    for i, b in enumerate(host):
        b["R"] = b["R"] * (1.0 - 0.1 * i)
    losses = []
    for mode in ("resident", "prefetch"):
        model, p = build_model(cfg, params, weights=z)
        src = DevicePrefetcher(host, DEV) if mode == "prefetch" else [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()} for b in host]
        cur = []
        for b in src:
            loss = step_forward(model, b, p)[0]
            loss.backward()
            cur.append(float(loss))
        losses.append(cur)
    assert np.allclose(losses[0], losses[1], rtol=0, atol=1e-6), losses


def test_refuses_cpu_and_single_slot():
    with pytest.raises(RuntimeError):
        DevicePrefetcher([], "cpu")
    with pytest.raises(ValueError):
        DevicePrefetcher([], DEV, depth=1)
