"""Thin torch-tensor wrappers over the kernel launchers of libcrct_hip.so.

torch is used only for device memory and the current HIP stream; every computation below is one
C-ABI call into the hand-written gfx950 kernels (include/crct_hip.h).  bf16 tensors are passed as
``torch.bfloat16``; all tensors must be contiguous CUDA tensors.  These wrappers exist for the kernel
parity tests and for building blocks outside the step engine; the training step itself goes
through ``crct.engine`` (one native call per forward / backward).
"""
import ctypes as C

import torch

from . import lib as L

ACT = dict(none=0, gelu=1, relu=2, leaky=3, tanh=4)


def _chk(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libcrct_hip kernels need CUDA (HIP) tensors; got a %s tensor -- no CPU fallback" % t.device)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("expected %s, got %s" % (dtype, t.dtype))
    return t


def _drop(p, site):
    thr = L.drop_threshold(p)
    return thr, (1.0 / (1.0 - p) if thr else 1.0), int(site)


def _gemm_args(g, A, B, M, N, K, ta=False, tb=False, lda=None, ldb=None, out=None, ldc=None, bias=None, act="none",
               preact_out=None, dact_src=None, dact="none", ld_aux=None, addend=None, ld_add=None, out_f32=False,
               accumulate=False, tile=-1, alpha=1.0, p_drop=0.0, site=0, seed=0, rowsum_out=None, split_k=0, model_site=0, c_cached=False):
    """addend may be bf16 or fp32 (the fp32 residual stream: CrctGemmArgs.addend_f32); c_cached: an fp32 output that the next kernel
    reads (ordinary instead of streaming stores)."""
    _chk(A, torch.bfloat16), _chk(B, torch.bfloat16)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    g.A, g.B, g.C = L.ptr(A), L.ptr(B), L.ptr(out)
    g.bias, g.preact_out, g.dact_src, g.addend = L.ptr(bias), L.ptr(preact_out), L.ptr(dact_src), L.ptr(addend)
    g.lda = lda if lda is not None else (M if ta else K)
    g.ldb = ldb if ldb is not None else (N if tb else K)
    g.ldc = ldc if ldc is not None else N
    g.ld_aux = ld_aux if ld_aux is not None else N
    g.ld_add = ld_add if ld_add is not None else N
    g.M, g.N, g.K, g.ta, g.tb = M, N, K, int(ta), int(tb)
    g.act, g.dact, g.c_is_f32, g.accumulate, g.tile, g.alpha = ACT[act], ACT[dact], int(out.dtype == torch.float32), int(accumulate), tile, alpha
    g.drop_thr, g.drop_scale, g.drop_site = _drop(p_drop, site)
    g.seed = seed
    g.rowsum_out = L.ptr(rowsum_out)
    g.site = int(model_site)
    g.addend_f32 = int(addend is not None and addend.dtype == torch.float32)
    g.c_cached = int(bool(c_cached))
    if split_k and split_k > 1:       # K-partitioned launch: slab space + ticket words (zero before the first use) per device
        ws, cnt = _splitk_space(A.device, M, N, split_k)
        g.split_k, g.splitk_ws, g.splitk_cnt = int(split_k), L.ptr(ws), L.ptr(cnt)
        g._keep = (ws, cnt)
    return out


_SPLITK = {}


def _splitk_space(dev, M, N, S):
    lib = L.load()
    need, tickets = lib.crct_gemm_splitk_ws_elems(M, N, S), lib.crct_gemm_splitk_tickets(M, N)
    ws, cnt = _SPLITK.get(dev, (None, None))
    if ws is None or ws.numel() < need or cnt.numel() < tickets:
        ws = torch.empty(max(need, ws.numel() if ws is not None else 0), dtype=torch.float32, device=dev)
        cnt = torch.zeros(max(tickets, 4096), dtype=torch.int32, device=dev)
        _SPLITK[dev] = (ws, cnt)
    return ws, cnt


def gemm(A, B, M, N, K, **kw):
    """One GEMM with its fused epilogue (keyword arguments: see ``_gemm_args``)."""
    g = L.GemmArgs()
    out = _gemm_args(g, A, B, M, N, K, **kw)
    L.check(L.load().crct_gemm_bf16(C.byref(g), L.current_stream()), "gemm")
    return out


def gemm_grouped(problems):
    """``problems`` = [dict(A=, B=, M=, N=, K=, ...gemm keywords)] sharing (ta, tb): ONE grouped launch (crct_gemm_bf16_grouped),
    every problem with its own epilogue.  Returns the outputs."""
    arr = (L.GemmArgs * len(problems))()
    outs = [_gemm_args(g, **prob) for g, prob in zip(arr, problems)]
    L.check(L.load().crct_gemm_bf16_grouped(arr, len(problems), L.current_stream()), "gemm_grouped")
    return outs


def gemm_fp8(Aq, Bq, scale_a, scale_b, M, N, K, bias=None, act="none", preact_out=None, addend=None, p_drop=0.0, site=0, seed=0,
             out_f32=False, q_out=None, q_scale=None, q_amax=None, a_bf8=False, q_bf8=False, dact_src=None, dact="none"):
    """y = act((Aq / sa) (Bq / sb)^T + bias) from OCP e4m3 operands (uint8 / float8_e4m3fn tensors [M][K], [N][K]); scale_* are
    device fp32 scalars.  q_out: e4m3 copy of y quantised with q_scale, max |y| max-ed into q_amax.  a_bf8: Aq holds OCP e5m2
    (a gradient: the fp8 data-gradient GEMMs); q_bf8: the q_out copy is e5m2."""
    lib = L.load()
    out = torch.empty(M, N, device=Aq.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    g = L.GemmArgs()
    g.A, g.B, g.C = L.ptr(Aq), L.ptr(Bq), L.ptr(out)
    g.bias, g.preact_out, g.addend = L.ptr(bias), L.ptr(preact_out), L.ptr(addend)
    g.lda, g.ldb, g.ldc, g.ld_aux, g.ld_add = K, K, N, N, N
    g.M, g.N, g.K = M, N, K
    g.act, g.c_is_f32, g.tile, g.alpha = ACT[act], int(out_f32), -1, 1.0
    g.drop_thr, g.drop_scale, g.drop_site = _drop(p_drop, site)
    g.seed = seed
    g.dact_src, g.dact = L.ptr(dact_src), ACT[dact]
    g.fp8, g.scale_a, g.scale_b = 1 | (2 if a_bf8 else 0) | (4 if q_bf8 else 0), L.ptr(scale_a), L.ptr(scale_b)
    g.q_out, g.q_scale, g.q_amax, g.ld_q = L.ptr(q_out), L.ptr(q_scale), L.ptr(q_amax), N
    L.check(lib.crct_gemm_bf16(C.byref(g), L.current_stream()), "gemm_fp8")
    return out


def gemm_wgrad_grouped(problems):
    """``problems`` = [(dy[R][N], x[R][K], out[N][K] fp32[, db[N] fp32])]: out += dy^T x (db += colsum dy), ONE launch."""
    lib = L.load()
    arr = (L.GemmArgs * len(problems))()
    for g, prob in zip(arr, problems):
        dy, x, out = prob[:3]
        g.rowsum_out = L.ptr(prob[3]) if len(prob) > 3 else None      # optional bias gradient [N] += column sums of dy
        _chk(dy, torch.bfloat16), _chk(x, torch.bfloat16), _chk(out, torch.float32)
        R, N = dy.shape
        K = x.shape[1]
        g.A, g.B, g.C = L.ptr(dy), L.ptr(x), L.ptr(out)
        g.lda, g.ldb, g.ldc, g.ld_aux, g.ld_add = N, K, K, K, K
        g.M, g.N, g.K, g.ta, g.tb = N, K, R, 1, 1
        g.c_is_f32, g.accumulate, g.tile, g.alpha = 1, 1, -1, 1.0
    L.check(lib.crct_gemm_bf16_grouped(arr, len(problems), L.current_stream()), "gemm_grouped")


def gemm_wgrad_fp8(problems, accumulate=True, tile=-1):
    """fp8 weight gradients: ``problems`` = [(dyq[R][N] e5m2, xq[R][K] e4m3, scale_dy, scale_x, out[N][K] fp32)]:
    out (+)= (dyq / s_dy)^T (xq / s_x) with fp32 accumulation; one launch for two or more problems, R (tokens) arbitrary,
    N and K multiples of 16."""
    lib = L.load()
    arr = (L.GemmArgs * len(problems))()
    for g, (dyq, xq, s_dy, s_x, out) in zip(arr, problems):
        _chk(out, torch.float32)
        R, N = dyq.shape
        K = xq.shape[1]
        assert xq.shape[0] == R and out.shape == (N, K) and dyq.element_size() == 1 and xq.element_size() == 1
        g.A, g.B, g.C = L.ptr(dyq), L.ptr(xq), L.ptr(out)
        g.lda, g.ldb, g.ldc, g.ld_aux, g.ld_add = N, K, K, K, K
        g.M, g.N, g.K, g.ta, g.tb = N, K, R, 1, 1
        g.c_is_f32, g.accumulate, g.tile, g.alpha = 1, int(accumulate), tile, 1.0
        g.fp8, g.scale_a, g.scale_b = 1 | 2, L.ptr(s_dy), L.ptr(s_x)
    if len(problems) == 1:
        L.check(lib.crct_gemm_bf16(C.byref(arr[0]), L.current_stream()), "gemm_wgrad_fp8")
    else:
        L.check(lib.crct_gemm_bf16_grouped(arr, len(problems), L.current_stream()), "gemm_wgrad_fp8 (grouped)")


def layernorm_fwd(x, gamma, beta, eps=1e-12, p_drop=0.0, site=0, seed=0, y_f32=False):
    """x bf16 or fp32 [M, H] (fp32: the pre-LayerNorm sums of the fp32 residual stream); y_f32: also return y as fp32 (4th result)."""
    lib = L.load()
    M, H = x.shape
    y = torch.empty(M, H, device=x.device, dtype=torch.bfloat16)
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    thr, sc, st = _drop(p_drop, site)
    if x.dtype == torch.float32 or y_f32:
        y32 = torch.empty(M, H, device=x.device, dtype=torch.float32) if y_f32 else None
        a = L.LnFwdArgs()
        a.x, a.gamma, a.beta, a.y, a.mean, a.rstd = L.ptr(_chk(x)), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd)
        a.M, a.H, a.eps, a.drop_thr, a.drop_scale, a.drop_site, a.seed = M, H, eps, thr, sc, st, seed
        a.x_f32, a.y_f32 = int(x.dtype == torch.float32), L.ptr(y32)
        L.check(lib.crct_layernorm_fwd_args(C.byref(a), L.current_stream()), "layernorm_fwd")
        return (y, mean, rstd, y32) if y_f32 else (y, mean, rstd)
    L.check(lib.crct_layernorm_fwd(L.ptr(_chk(x, torch.bfloat16)), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                   M, H, eps, thr, sc, st, seed, L.current_stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, want_lin=False, p_lin=0.0, lin_site=0, p_post=0.0, post_site=0, seed=0,
                  dgamma=None, dbeta=None, dbias=None, accumulate=False):
    lib = L.load()
    M, H = x.shape
    dx = torch.empty_like(dy)
    dxl = torch.empty_like(dy) if want_lin else None
    nb = lib.crct_layernorm_bwd_blocks(M)
    part = torch.empty(3 * 4 * nb * H, device=x.device, dtype=torch.float32)       # one partial row per wave
    dgamma = torch.zeros(H, device=x.device) if dgamma is None else dgamma
    dbeta = torch.zeros(H, device=x.device) if dbeta is None else dbeta
    dbias = torch.zeros(H, device=x.device) if dbias is None else dbias
    pt, ps, psite = _drop(p_post, post_site)
    lt, ls, lsite = _drop(p_lin, lin_site)
    if x.dtype == torch.float32:      # the saved pre-norm rows of the fp32 residual stream: rows pass from the struct, then the column pass
        a = L.LnBwdArgs()
        a.dy, a.x, a.mean, a.rstd, a.gamma, a.dx, a.dx_lin, a.partials = (L.ptr(_chk(dy, torch.bfloat16)), L.ptr(x), L.ptr(mean), L.ptr(rstd),
                                                                            L.ptr(gamma), L.ptr(dx), L.ptr(dxl), L.ptr(part))
        a.M, a.H, a.post_thr, a.post_scale, a.post_site, a.lin_thr, a.lin_scale, a.lin_site, a.seed = M, H, pt, ps, psite, lt, ls, lsite, seed
        a.x_f32 = 1
        L.check(lib.crct_layernorm_bwd_rows_args(C.byref(a), L.current_stream()), "layernorm_bwd")
        L.check(lib.crct_layernorm_bwd_finalize(L.ptr(part), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), M, H, int(accumulate), L.current_stream()),
                "layernorm_bwd (finalize)")
        return dx, dxl, dgamma, dbeta, dbias
    L.check(lib.crct_layernorm_bwd(L.ptr(_chk(dy, torch.bfloat16)), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(dx), L.ptr(dxl),
                                   L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), L.ptr(part), M, H, int(accumulate),
                                   pt, ps, psite, lt, ls, lsite, seed, L.current_stream()), "layernorm_bwd")
    return dx, dxl, dgamma, dbeta, dbias


def colsum(x, M, N, ld=None, out=None, accumulate=False):
    lib = L.load()
    nb = lib.crct_colsum_blocks(M)
    part = torch.empty(nb * N, device=x.device, dtype=torch.float32)
    out = torch.zeros(N, device=x.device) if out is None else out
    L.check(lib.crct_colsum_bf16(L.ptr(_chk(x, torch.bfloat16)), ld if ld is not None else N, L.ptr(out), L.ptr(part), M, N,
                                 int(accumulate), L.current_stream()), "colsum")
    return out


def softmax_rows(x):
    lib = L.load()
    M, F = x.shape
    y = torch.empty(M, F, device=x.device, dtype=torch.bfloat16)
    L.check(lib.crct_softmax_rows_f32_bf16(L.ptr(_chk(x, torch.float32)), L.ptr(y), M, F, L.current_stream()), "softmax_rows")
    return y


def cast_bf16(x, out=None):
    lib = L.load()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out is None else out
    L.check(lib.crct_cast_f32_bf16(L.ptr(_chk(x, torch.float32)), L.ptr(out), x.numel(), L.current_stream()), "cast")
    return out


def attention_fwd(q, k, v, keymask, heads, d, p_drop=0.0, site=0, seed=0, row_lse=None):
    """q [B,Tq,ldq] / k,v [B,Tk,ld*] bf16 (may be column slices of a wider buffer: pass the *slice*).
    row_lse: fp32 [B, heads, Tq] that the long-sequence kernels fill with the softmax row statistics (CrctAttnQuant.row_lse; the
    other kernels leave it untouched) -- hand it, with the returned ctx, to ``attention_bwd``."""
    lib = L.load()
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[1]
    ctx = torch.empty(B, Tq, heads * d, device=q.device, dtype=torch.bfloat16)
    thr, sc, st = _drop(p_drop, site)
    if row_lse is not None:
        assert tuple(row_lse.shape) == (B, heads, Tq)
        qz = L.AttnQuant()
        qz.row_lse = L.ptr(_chk(row_lse, torch.float32))
        L.check(lib.crct_attention_fwd_q(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(_chk(keymask, torch.uint8)), L.ptr(ctx), B, heads, Tq, Tk, d,
                                         q.stride(1), k.stride(1), v.stride(1), heads * d, thr, sc, st, seed, C.byref(qz), L.current_stream()),
                "attention_fwd")
        return ctx
    L.check(lib.crct_attention_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(_chk(keymask, torch.uint8)), L.ptr(ctx), B, heads, Tq, Tk, d,
                                   q.stride(1), k.stride(1), v.stride(1), heads * d, thr, sc, st, seed, L.current_stream()),
            "attention_fwd")
    return ctx


def attention_fwd_q(q, k, v, keymask, heads, d, q_scale, q_amax, p_drop=0.0, site=0, seed=0):
    """attention_fwd that also returns the e4m3 copy of ctx (uint8, same shape), quantised with the device scalar q_scale."""
    lib = L.load()
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[1]
    ctx = torch.empty(B, Tq, heads * d, device=q.device, dtype=torch.bfloat16)
    ctx_q = torch.zeros(B, Tq, heads * d, device=q.device, dtype=torch.uint8)
    qz = L.AttnQuant()
    qz.ctx_q, qz.ctx_scale, qz.ctx_amax = L.ptr(ctx_q), L.ptr(q_scale), L.ptr(q_amax)
    thr, sc, st = _drop(p_drop, site)
    L.check(lib.crct_attention_fwd_q(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(_chk(keymask, torch.uint8)), L.ptr(ctx), B, heads, Tq, Tk, d,
                                     q.stride(1), k.stride(1), v.stride(1), heads * d, thr, sc, st, seed, C.byref(qz), L.current_stream()),
            "attention_fwd_q")
    return ctx, ctx_q


def attention_bwd_q(q, k, v, keymask, dctx, heads, d, dq_scale, dq_amax, dkv_scale, dkv_amax, p_drop=0.0, site=0, seed=0, row_lse=None,
                    ctx=None):
    """attention_bwd that also returns the e5m2 copies of dq, dk, dv (uint8); row_lse / ctx as in ``attention_bwd``."""
    lib = L.load()
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[1]
    dq = torch.empty(B, Tq, heads * d, device=q.device, dtype=torch.bfloat16)
    dk = torch.empty(B, Tk, heads * d, device=q.device, dtype=torch.bfloat16)
    dv = torch.empty_like(dk)
    dq8, dk8, dv8 = (torch.zeros(t.shape, device=q.device, dtype=torch.uint8) for t in (dq, dk, dv))
    qz = L.AttnQuant()
    qz.dq_q, qz.dk_q, qz.dv_q = L.ptr(dq8), L.ptr(dk8), L.ptr(dv8)
    qz.dq_scale, qz.dq_amax, qz.dkv_scale, qz.dkv_amax = L.ptr(dq_scale), L.ptr(dq_amax), L.ptr(dkv_scale), L.ptr(dkv_amax)
    if row_lse is not None or ctx is not None:
        qz.row_lse, qz.ctx, qz.ld_ctx = L.ptr(_chk(row_lse, torch.float32)), L.ptr(_chk(ctx, torch.bfloat16)), (ctx.stride(1) if ctx is not None else 0)
    thr, sc, st = _drop(p_drop, site)
    L.check(lib.crct_attention_bwd_q(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(keymask), L.ptr(_chk(dctx, torch.bfloat16)), L.ptr(dq), L.ptr(dk), L.ptr(dv),
                                     B, heads, Tq, Tk, d, q.stride(1), k.stride(1), v.stride(1), dctx.stride(1),
                                     heads * d, heads * d, heads * d, thr, sc, st, seed, C.byref(qz), L.current_stream()), "attention_bwd_q")
    return (dq, dk, dv), (dq8, dk8, dv8)


def attention_bwd(q, k, v, keymask, dctx, heads, d, p_drop=0.0, site=0, seed=0, row_lse=None, ctx=None):
    """row_lse / ctx: what ``attention_fwd(..., row_lse=)`` of the same operands produced -- the long-sequence kernels then skip
    their statistics sweep (the other kernels ignore both)."""
    lib = L.load()
    B, Tq = q.shape[0], q.shape[1]
    Tk = k.shape[1]
    dq = torch.empty(B, Tq, heads * d, device=q.device, dtype=torch.bfloat16)
    dk = torch.empty(B, Tk, heads * d, device=q.device, dtype=torch.bfloat16)
    dv = torch.empty_like(dk)
    thr, sc, st = _drop(p_drop, site)
    if row_lse is not None or ctx is not None:
        qz = L.AttnQuant()
        qz.row_lse, qz.ctx, qz.ld_ctx = L.ptr(_chk(row_lse, torch.float32)), L.ptr(_chk(ctx, torch.bfloat16)), (ctx.stride(1) if ctx is not None else 0)
        L.check(lib.crct_attention_bwd_q(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(keymask), L.ptr(_chk(dctx, torch.bfloat16)), L.ptr(dq), L.ptr(dk),
                                         L.ptr(dv), B, heads, Tq, Tk, d, q.stride(1), k.stride(1), v.stride(1), dctx.stride(1),
                                         heads * d, heads * d, heads * d, thr, sc, st, seed, C.byref(qz), L.current_stream()), "attention_bwd")
        return dq, dk, dv
    L.check(lib.crct_attention_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(keymask), L.ptr(_chk(dctx, torch.bfloat16)), L.ptr(dq), L.ptr(dk), L.ptr(dv),
                                   B, heads, Tq, Tk, d, q.stride(1), k.stride(1), v.stride(1), dctx.stride(1),
                                   heads * d, heads * d, heads * d, thr, sc, st, seed, L.current_stream()), "attention_bwd")
    return dq, dk, dv


def adamw_plan(seg_len):
    """Host-side block table for crct_adamw_step: returns (blk_seg int32[n], blk_off int64[n])."""
    lib = L.load()
    lens = torch.as_tensor(seg_len, dtype=torch.int64)
    n = lib.crct_adamw_plan(lens.data_ptr(), lens.numel(), None, None, 0)
    blk_seg = torch.empty(n, dtype=torch.int32)
    blk_off = torch.empty(n, dtype=torch.int64)
    lib.crct_adamw_plan(lens.data_ptr(), lens.numel(), blk_seg.data_ptr(), blk_off.data_ptr(), n)
    return blk_seg, blk_off
