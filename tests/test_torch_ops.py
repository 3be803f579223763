"""``torch.ops.crct.*`` (crct/torch_ops.py): the registered custom-op surface north_star names.

not gpu: the ops exist with the documented schemas and there is NO CPU implementation behind them (a CPU call fails in the
dispatcher instead of falling back).  gpu: the differentiable building blocks against plain fp32 PyTorch on the same
bf16-rounded operands (tolerances of tests/test_kernels_gpu.py: 1e-2 of max for bf16 results, 2e-3 for fp32), and the training
step really goes through ``crct::step_forward`` / ``crct::step_backward``.
"""
import math

import pytest
import torch

from crct import torch_ops as T


def test_ops_are_registered_and_have_no_cpu_path():
    for name in T.OP_NAMES:
        assert hasattr(torch.ops.crct, name), name
    x, w = torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        torch.ops.crct.linear_fwd(x, w)
    with pytest.raises(NotImplementedError):
        torch.ops.crct.linear(x, w, None)
    with pytest.raises(NotImplementedError):
        torch.ops.crct.layernorm(x, torch.ones(64), torch.zeros(64))
    with pytest.raises(NotImplementedError):
        torch.ops.crct.step_forward(0, torch.zeros(1), torch.zeros(1), [None])
    schema = str(torch.ops.crct.step_backward.default._schema)
    assert "Tensor(a!) grads" in schema and "Tensor?[] batch" in schema


def _rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.gpu
def test_linear_layernorm_attention_ops_differentiate_like_torch():
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    B, Tn, H, N, heads = 5, 20, 768, 1024, 12
    d = H // heads
    x = torch.randn(B, Tn, H, generator=g).to(dev).bfloat16().requires_grad_()
    w = (torch.randn(N, H, generator=g) * 0.03).to(dev).bfloat16().requires_grad_()
    b = torch.randn(N, generator=g).to(dev).requires_grad_()
    y = torch.ops.crct.linear(x, w, b)
    up = torch.randn(B, Tn, N, generator=g).to(dev).bfloat16()
    y.backward(up)
    xr, wr, br = x.detach().float().requires_grad_(), w.detach().float().requires_grad_(), b.detach().clone().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(up.float())
    assert _rel(y, yr) < 1e-2 and _rel(x.grad, xr.grad) < 1e-2 and _rel(w.grad, wr.grad) < 1e-2 and _rel(b.grad, br.grad) < 1e-2

    gamma, beta = (torch.rand(H, generator=g) + 0.5).to(dev).requires_grad_(), torch.randn(H, generator=g).to(dev).requires_grad_()
    x2 = torch.randn(B, Tn, H, generator=g).to(dev).bfloat16().requires_grad_()
    z = torch.ops.crct.layernorm(x2, gamma, beta, 1e-12)
    upz = torch.randn(B, Tn, H, generator=g).to(dev).bfloat16()
    z.backward(upz)
    x2r, gr, btr = x2.detach().float().requires_grad_(), gamma.detach().clone().requires_grad_(), beta.detach().clone().requires_grad_()
    zr = torch.nn.functional.layer_norm(x2r, (H,), gr, btr, 1e-12)
    zr.backward(upz.float())
    assert _rel(z, zr) < 1e-2 and _rel(x2.grad, x2r.grad) < 1.5e-2 and _rel(gamma.grad, gr.grad) < 2e-3 and _rel(beta.grad, btr.grad) < 2e-3

    q, k, v = (torch.randn(B, Tn, H, generator=g).to(dev).bfloat16().requires_grad_() for _ in range(3))
    mask = torch.ones(B, Tn, dtype=torch.uint8, device=dev)
    mask[:, Tn - 4:] = 0
    ctx = torch.ops.crct.attention(q, k, v, mask, heads, d)
    upc = torch.randn(B, Tn, H, generator=g).to(dev).bfloat16()
    ctx.backward(upc)

    def split(t):
        return t.view(B, Tn, heads, d).permute(0, 2, 1, 3)
    qr, kr, vr = (t.detach().float().requires_grad_() for t in (q, k, v))
    s = split(qr) @ split(kr).transpose(-1, -2) / math.sqrt(d) + (1.0 - mask.float())[:, None, None, :] * -10000.0
    cr = (torch.softmax(s, -1) @ split(vr)).permute(0, 2, 1, 3).reshape(B, Tn, H)
    cr.backward(upc.float())
    assert _rel(ctx, cr) < 1e-2
    for a, r in ((q, qr), (k, kr), (v, vr)):
        assert _rel(a.grad, r.grad) < 2e-2


@pytest.mark.gpu
def test_training_step_runs_through_the_registered_step_ops(monkeypatch):
    from crct import config as C, synthetic as S
    from crct.engine import StepEngine
    from crct.model import VisualDialogEncoder
    from crct.step_adapter import forward
    seen = []
    orig_f, orig_b = T._step_forward, T._step_backward     # the implementations registered for the CUDA key
    native_f, native_b = StepEngine.forward_native, StepEngine.backward_native
    # the ONLY callers of the native entry points are the op implementations: count there
    monkeypatch.setattr(StepEngine, "forward_native", lambda self, *a: (seen.append("fwd"), native_f(self, *a))[1])
    monkeypatch.setattr(StepEngine, "backward_native", lambda self, *a: (seen.append("bwd"), native_b(self, *a))[1])
    assert orig_f is not None and orig_b is not None
    cfg = C.tiny_config()
    params = dict(C.default_params(categories=9), device=torch.device("cuda:0"))
    model = VisualDialogEncoder(params, config=cfg)
    batch = S.make_batch(8, 9, 6, cfg.v_feature_size, categories=9, vocab_size=cfg.vocab_size, seed=5)
    out = forward(model, batch, params)
    out[0].backward()
    torch.cuda.synchronize()
    core = model.bert_pretrained
    eng = core._engine
    assert eng._op_handle is not None and T._ENGINES.get(eng._op_handle) is eng
    assert "fwd" in seen and "bwd" in seen
    # the ops called directly on the same engine and batch (evaluation configuration: no dropout): repeatable, in place
    tensors, step = eng._keep
    eng._staged = dict(step, training=False)
    a = torch.ops.crct.step_forward(eng._op_handle, core.flat_params, core.flat_shadow, T.pack_batch(tensors)).clone()
    b = torch.ops.crct.step_forward(eng._op_handle, core.flat_params, core.flat_shadow, T.pack_batch(tensors))
    torch.cuda.synchronize()
    assert b.data_ptr() == eng.out.data_ptr() and torch.equal(a, b)
    with pytest.raises(RuntimeError):
        torch.ops.crct.step_forward(12345, core.flat_params, core.flat_shadow, T.pack_batch(tensors))
