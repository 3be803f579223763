"""``torch.ops.crct.*`` -- the HIP kernels and the step engine as registered custom torch ops (BASELINE north_star:
"exposing the kernels as custom torch ops").

Registered with ``torch.library`` for the CUDA (= HIP) dispatch key ONLY: there is no CPU implementation, so an op
called on CPU tensors fails in the dispatcher (``NotImplementedError``), and without ``libcrct_hip.so`` the first call
raises from ``crct.lib.load()``.  Every op runs on the current HIP stream and does no host synchronisation.

Kernel-level ops (functional, one C-ABI launch each; ``crct/ops.py`` holds the argument marshalling):

  crct::linear_fwd(x, w, bias?, act)            y = act(x w^T + b)           vilbert.py:393-395, 459-489 ...
  crct::linear_dgrad(dy, w)                     dx = dy w
  crct::linear_wgrad(dy, x)                     dw = dy^T x   (fp32)
  crct::layernorm_fwd(x, gamma, beta, eps)      -> (y, mean, rstd)           vilbert.py:281-294
  crct::layernorm_bwd(dy, x, mean, rstd, gamma) -> (dx, dgamma, dbeta)
  crct::attention_fwd(q, k, v, keymask, heads, d)                -> ctx      vilbert.py:396-447
  crct::attention_bwd(q, k, v, keymask, dctx, heads, d)          -> (dq, dk, dv)

Differentiable ops built from them (``torch.library.register_autograd``): ``crct::linear``, ``crct::layernorm``,
``crct::attention`` -- usable as drop-in building blocks inside an ordinary ``nn.Module``.

Step ops (what ``crct.model`` calls; one native engine call each, ``include/crct_hip.h`` crct_engine_forward / _backward):

  crct::step_forward(engine, params, shadow, batch[]) -> out      out = [24 stats | 5*B regressed values | logits]
  crct::step_backward(engine, params, shadow, grads, batch[], seg) -> ()

``engine`` is the integer handle of a live ``crct.engine.StepEngine`` (``engine_handle()``); the step configuration
(loss kind, coefficients, seed, events, fp8 state: scalars and raw handles, not tensors) is staged on the engine object
by ``StepEngine.forward()`` / ``backward()`` (the callers of these two ops) right before the op.
"""
import weakref

import torch

from . import ops

_LIB = torch.library.Library("crct", "DEF")
_ENGINES = weakref.WeakValueDictionary()

BATCH_KEYS = ("tokens", "segments", "loc", "image_feat", "image_loc", "image_target", "R", "labels", "sep_indices", "hist_len",
              "image_mask", "text_keymask", "image_keymask")


def engine_handle(eng):
    """Register a StepEngine and return the integer the step ops take."""
    key = id(eng)
    _ENGINES[key] = eng
    return key


def pack_batch(tensors):
    """The batch dict as the positional ``Tensor?[]`` of the step ops (order = BATCH_KEYS)."""
    return [tensors.get(k) for k in BATCH_KEYS]


def _unpack_batch(batch):
    return {k: t for k, t in zip(BATCH_KEYS, batch) if t is not None}


def _engine(handle):
    eng = _ENGINES.get(handle)
    if eng is None:
        raise RuntimeError("crct::step_*: %d is not a live StepEngine handle" % handle)
    return eng


_LIB.define("linear_fwd(Tensor x, Tensor w, Tensor? bias=None, str act='none') -> Tensor")
_LIB.define("linear_dgrad(Tensor dy, Tensor w) -> Tensor")
_LIB.define("linear_wgrad(Tensor dy, Tensor x) -> Tensor")
_LIB.define("layernorm_fwd(Tensor x, Tensor gamma, Tensor beta, float eps=1e-12) -> (Tensor, Tensor, Tensor)")
_LIB.define("layernorm_bwd(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor gamma) -> (Tensor, Tensor, Tensor)")
_LIB.define("attention_fwd(Tensor q, Tensor k, Tensor v, Tensor keymask, int heads, int d) -> Tensor")
_LIB.define("attention_bwd(Tensor q, Tensor k, Tensor v, Tensor keymask, Tensor dctx, int heads, int d) -> (Tensor, Tensor, Tensor)")
_LIB.define("linear(Tensor x, Tensor w, Tensor? bias=None) -> Tensor")
_LIB.define("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps=1e-12) -> Tensor")
_LIB.define("attention(Tensor q, Tensor k, Tensor v, Tensor keymask, int heads, int d) -> Tensor")
_LIB.define("step_forward(int engine, Tensor params, Tensor shadow, Tensor?[] batch) -> Tensor")
_LIB.define("step_backward(int engine, Tensor params, Tensor shadow, Tensor(a!) grads, Tensor?[] batch, int seg=-1) -> ()")


def _rows(x):
    if x.dim() < 2:
        raise RuntimeError("crct ops take [..., features] tensors; got shape %s" % (tuple(x.shape),))
    return x.reshape(-1, x.shape[-1])


def _bf16(t, name):
    if t.dtype != torch.bfloat16:
        raise RuntimeError("crct op: %s must be bfloat16 (bf16 storage, fp32 accumulation); got %s" % (name, t.dtype))
    return t.contiguous()


def _linear_fwd(x, w, bias=None, act="none"):
    x2, w = _bf16(_rows(x), "x"), _bf16(w, "w")
    if w.dim() != 2 or w.shape[1] != x2.shape[1]:
        raise RuntimeError("crct::linear_fwd: weight %s does not match input features %d" % (tuple(w.shape), x2.shape[1]))
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != w.shape[0]):
        raise RuntimeError("crct::linear_fwd: bias must be float32 [%d]" % w.shape[0])
    y = ops.gemm(x2, w, x2.shape[0], w.shape[0], x2.shape[1], bias=bias, act=act)
    return y.view(*x.shape[:-1], w.shape[0])


def _linear_dgrad(dy, w):
    d2, w = _bf16(_rows(dy), "dy"), _bf16(w, "w")
    if w.shape[0] != d2.shape[1]:
        raise RuntimeError("crct::linear_dgrad: dy features %d != weight rows %d" % (d2.shape[1], w.shape[0]))
    dx = ops.gemm(d2, w, d2.shape[0], w.shape[1], w.shape[0], tb=True)
    return dx.view(*dy.shape[:-1], w.shape[1])


def _linear_wgrad(dy, x):
    d2, x2 = _bf16(_rows(dy), "dy"), _bf16(_rows(x), "x")
    if d2.shape[0] != x2.shape[0]:
        raise RuntimeError("crct::linear_wgrad: %d gradient rows vs %d input rows" % (d2.shape[0], x2.shape[0]))
    return ops.gemm(d2, x2, d2.shape[1], x2.shape[1], d2.shape[0], ta=True, tb=True, out_f32=True)


def _layernorm_fwd(x, gamma, beta, eps=1e-12):
    x2 = _bf16(_rows(x), "x")
    y, mean, rstd = ops.layernorm_fwd(x2, gamma.float().contiguous(), beta.float().contiguous(), eps=eps)
    return y.view(x.shape), mean, rstd


def _layernorm_bwd(dy, x, mean, rstd, gamma):
    d2, x2 = _bf16(_rows(dy), "dy"), _bf16(_rows(x), "x")
    dx, _, dg, db, _ = ops.layernorm_bwd(d2, x2, mean, rstd, gamma.float().contiguous())
    return dx.view(x.shape), dg, db


def _attention_fwd(q, k, v, keymask, heads, d):
    return ops.attention_fwd(_bf16(q, "q"), _bf16(k, "k"), _bf16(v, "v"), keymask.to(torch.uint8).contiguous(), heads, d)


def _attention_bwd(q, k, v, keymask, dctx, heads, d):
    return ops.attention_bwd(_bf16(q, "q"), _bf16(k, "k"), _bf16(v, "v"), keymask.to(torch.uint8).contiguous(), _bf16(dctx, "dctx"),
                             heads, d)


def _step_forward(engine, params, shadow, batch):
    eng = _engine(engine)
    tensors, step = _unpack_batch(batch), eng.staged_step()
    eng.forward_native(params, shadow, tensors, step)
    return eng.out


def _step_backward(engine, params, shadow, grads, batch, seg=-1):
    eng = _engine(engine)
    eng.backward_native(params, shadow, grads, _unpack_batch(batch), eng.staged_step(), seg)


for _name, _fn in (("linear_fwd", _linear_fwd), ("linear_dgrad", _linear_dgrad), ("linear_wgrad", _linear_wgrad),
                   ("layernorm_fwd", _layernorm_fwd), ("layernorm_bwd", _layernorm_bwd), ("attention_fwd", _attention_fwd),
                   ("attention_bwd", _attention_bwd), ("step_forward", _step_forward), ("step_backward", _step_backward)):
    _LIB.impl(_name, _fn, "CUDA")


# ---- differentiable building blocks -------------------------------------------------------------------------------------
def _linear(x, w, bias=None):
    return torch.ops.crct.linear_fwd(x, w, bias, "none")


def _linear_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.crct.linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
    dw = torch.ops.crct.linear_wgrad(dy, x).to(w.dtype) if ctx.needs_input_grad[1] else None
    db = dy.reshape(-1, dy.shape[-1]).float().sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return dx, dw, db


def _layernorm(x, gamma, beta, eps=1e-12):
    return torch.ops.crct.layernorm_fwd(x, gamma, beta, eps)[0]


def _layernorm_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    # the statistics are recomputed by the forward op in backward (they are not outputs of the differentiable op)
    ctx.save_for_backward(x, gamma, beta)
    ctx.eps = eps


def _layernorm_backward(ctx, dy):
    x, gamma, beta = ctx.saved_tensors
    _, mean, rstd = torch.ops.crct.layernorm_fwd(x, gamma, beta, ctx.eps)
    dx, dg, db = torch.ops.crct.layernorm_bwd(dy.contiguous(), x, mean, rstd, gamma)
    return dx, dg.to(gamma.dtype), db.to(beta.dtype), None


def _attention(q, k, v, keymask, heads, d):
    return torch.ops.crct.attention_fwd(q, k, v, keymask, heads, d)


def _attention_setup(ctx, inputs, output):
    q, k, v, keymask, heads, d = inputs
    ctx.save_for_backward(q, k, v, keymask)
    ctx.heads, ctx.d = heads, d


def _attention_backward(ctx, dctx):
    q, k, v, keymask = ctx.saved_tensors
    dq, dk, dv = torch.ops.crct.attention_bwd(q, k, v, keymask, dctx.contiguous(), ctx.heads, ctx.d)
    return dq.view_as(q), dk.view_as(k), dv.view_as(v), None, None, None


_LIB.impl("linear", _linear, "CUDA")
_LIB.impl("layernorm", _layernorm, "CUDA")
_LIB.impl("attention", _attention, "CUDA")
torch.library.register_autograd("crct::linear", _linear_backward, setup_context=_linear_setup, lib=_LIB)
torch.library.register_autograd("crct::layernorm", _layernorm_backward, setup_context=_layernorm_setup, lib=_LIB)
torch.library.register_autograd("crct::attention", _attention_backward, setup_context=_attention_setup, lib=_LIB)

OP_NAMES = ("linear_fwd", "linear_dgrad", "linear_wgrad", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd",
            "linear", "layernorm", "attention", "step_forward", "step_backward")
