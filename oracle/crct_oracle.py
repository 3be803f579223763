"""CPU ORACLE for the CRCT co-attention training step  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, pure-PyTorch fp32 restatement of the reference hot path (levymsn/CQA-CRCT,
``CRCT/backbone/{vilbert,regressor,encoder_decorator}.py``), written against a flat
``{state_dict key: tensor}`` weight dict instead of the reference's nn.Module tree.  Backward is
torch autograd over these functions.  Each function cites the reference lines it restates.

Pinning: the reference holds no tests/golden vectors for this path, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, imported in the build container:
``tests/golden/make_golden.py`` runs the reference model and this oracle on identical weights and
inputs and commits the reference's outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
re-checks the oracle against those files wherever the tests run.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``cqa-crct_amd/``) never does.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # vilbert.py:282 (TF style: inside the sqrt)


# ----------------------------------------------------------------------------- primitives
def layer_norm(x, w, b):
    """vilbert.py:290-294 -- biased variance, eps inside the sqrt."""
    u = x.mean(-1, keepdim=True)
    s = ((x - u) ** 2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + LN_EPS)) + b


def gelu_erf(x):
    """vilbert.py:111-117."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


# fp8 emulation of the product's BASELINE configs[4] mode (test yardstick only; the reference has no fp8 mode): the forward of
# every encoder Linear whose two dimensions are multiples of 128 (QKV, attention output / biOutput, FFN) multiplies OCP e4m3
# roundings of the input and the weight (per-tensor scale 448 / max |.|), the backward uses the UNquantised operands -- the
# product with params['fp8_backward'] = False.  Switched on by ``FP8_EMULATION = True``.
FP8_EMULATION = False
# ... and of its fp8 BACKWARD (round 3): the data gradients dx = dy W of the same Linears multiply an OCP e5m2 rounding of dy
# (per-tensor scale 57344 / max |dy|; the product uses ONE scale for the fused dq | dk | dv gradient of a QKV projection, this
# emulation one per Linear) with the e4m3 rounding of the weight; the weight gradients use the unquantised operands.  Switched on
# by ``FP8_BWD_EMULATION = True`` (with FP8_EMULATION).
FP8_BWD_EMULATION = False
# ... and of the fp8 WEIGHT gradients of the same Linears: dW = dy^T x from the same e5m2 rounding of dy and the e4m3 rounding of x
# the forward GEMM read; the bias gradients use the unquantised dy.
# Switched on by ``FP8_WGRAD_EMULATION = True`` (with the two above).
FP8_WGRAD_EMULATION = False

# Block-scaled ("MX") variant of the same emulation (round 4, diagnostic): every operand of an fp8 GEMM is OCP e4m3 with one power-of-two
# (e8m0) scale per 32 consecutive elements ALONG THE CONTRACTION INDEX -- the operand format of gfx950's
# v_mfma_scale_f32_16x16x128_f8f6f4 -- gradients included (e4m3, not e5m2: the block scale carries the range).  ``FP8_MX = True``
# switches _Fp8Linear's roundings over; tools/lab/mx_emulation.py prices it against the per-tensor recipe.
FP8_MX = False
MX_BLOCK = 32
# ... and of the product's params['fp8_forward'] = False mode (round 4): the forward GEMMs multiply the unquantised operands, only the
# backward GEMMs (data and / or weight gradients, as the two flags above say) read fp8 roundings.  ``FP8_FWD_BF16 = True``.
FP8_FWD_BF16 = False


def _mx_round(t, dim):
    """e4m3 rounding of t with an e8m0 scale per MX_BLOCK elements along ``dim`` (scale = 2^ceil(log2(amax / 448)))."""
    x = t.movedim(dim, -1)
    shape = x.shape
    n = shape[-1]
    pad = (-n) % MX_BLOCK
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, MX_BLOCK)
    amax = xb.detach().abs().amax(-1, keepdim=True).clamp_min(2.0 ** -126)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (xb / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype) * scale
    q = q.reshape(*x.shape)[..., :n].reshape(shape)
    return q.movedim(-1, dim)


def _fp8_round(t, dim=-1):
    if FP8_MX:
        return _mx_round(t, dim)
    s = 448.0 / float(t.detach().abs().max().clamp_min(1e-30))
    return (t * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype) / s


def _bf8_round(t, dim=-1):
    if FP8_MX:
        return _mx_round(t, dim)
    s = 57344.0 / float(t.detach().abs().max().clamp_min(1e-30))
    return (t * s).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(t.dtype) / s


class _Fp8Linear(torch.autograd.Function):
    """fwd_q: the forward multiplies e4m3 roundings of x and w; bwd_q: the data gradient multiplies an e5m2 rounding of gy with the
    e4m3 rounding of w; wg_q: the weight gradient multiplies the e5m2 rounding of gy with the e4m3 rounding of x.  The bias
    gradient always sums the unquantised gy."""

    @staticmethod
    def forward(ctx, x, w, b, fwd_q=True, bwd_q=False, wg_q=False):
        ctx.save_for_backward(x, w)
        ctx.bwd_q, ctx.wg_q = bwd_q, wg_q
        return F.linear(_fp8_round(x), _fp8_round(w), b) if fwd_q else F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        # (block-scaled mode: the blocks run along each GEMM's own contraction index -- ``out`` for the data gradient, tokens for the
        # weight gradient -- so the two backward GEMMs read different roundings of gy)
        gq = _bf8_round(gy) if (ctx.bwd_q or ctx.wg_q) else None
        gx = (gq @ _fp8_round(w, 0)) if ctx.bwd_q else gy @ w
        if ctx.wg_q:
            g2 = gy.reshape(-1, gy.shape[-1])
            gw = (_bf8_round(g2, 0) if FP8_MX else gq.reshape(-1, gy.shape[-1])).t() @ _fp8_round(x.reshape(-1, x.shape[-1]), 0 if FP8_MX else -1)
        else:
            gw = gy.reshape(-1, gy.shape[-1]).t() @ x.reshape(-1, x.shape[-1])
        return gx, gw, gy.reshape(-1, gy.shape[-1]).sum(0), None, None, None


def _fp8_site(w):
    """The Linears of the encoder the product runs in fp8: both weight dimensions whole 128-deep fp8 K tiles (the forward
    contracts over ``in``, the data gradient over ``out``) -- QKV, attention output / biOutput, FFN up and down."""
    return w.shape[0] % 128 == 0 and w.shape[1] % 128 == 0


def linear(sd, prefix, x):
    w = sd[prefix + ".weight"]
    if FP8_EMULATION and ".encoder." in prefix:
        site = _fp8_site(w)
        fwd_q = site and not FP8_FWD_BF16
        bwd_q = FP8_BWD_EMULATION and site
        wg_q = FP8_WGRAD_EMULATION and bwd_q
        if site:
            return _Fp8Linear.apply(x, w, sd[prefix + ".bias"], fwd_q, bwd_q, wg_q)
    return F.linear(x, w, sd[prefix + ".bias"])


def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def _heads(x, n):
    B, L, H = x.shape
    return x.view(B, L, n, H // n).permute(0, 2, 1, 3)


def _merge(x):
    B, n, L, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, n * d)


def sdpa(q, k, v, add_mask, n_heads, p_drop, training):
    """softmax(q k^T / sqrt(d) + mask) v with prob-dropout (vilbert.py:392-412 / 522-543 / 684-723)."""
    qh, kh, vh = _heads(q, n_heads), _heads(k, n_heads), _heads(v, n_heads)
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(qh.shape[-1]) + add_mask
    p = _drop(torch.softmax(s, dim=-1), p_drop, training)
    return _merge(torch.matmul(p, vh))


# ----------------------------------------------------------------------------- embeddings
def embed_text(sd, cfg, input_ids, token_type_ids, loc, training):
    """BertEmbeddingLocation.forward, vilbert.py:320-358."""
    pre = "bert.embeddings."
    B, T = input_ids.shape
    not_qa = (token_type_ids != -1) & (token_type_ids != 1)
    pos = torch.arange(T, dtype=torch.long).unsqueeze(0).expand(B, T).clone()
    pos[not_qa] = T                                                    # :330
    pos = pos - pos.min(dim=-1)[0].unsqueeze(1)                        # :331
    pos[not_qa] = 0                                                    # :332
    pos_e = sd[pre + "position_embeddings.weight"][pos] * (~not_qa).unsqueeze(-1)    # :334-335
    word_e = sd[pre + "word_embeddings.weight"][input_ids]             # :336
    loc_e = linear(sd, pre + "txt_location_embeddings", loc)
    loc_e = loc_e * (loc.abs().sum(-1) != 0).unsqueeze(-1)             # :346-347
    tt = token_type_ids.clone()
    tt[tt == -1] = 0                                                   # :349-350
    type_e = sd[pre + "plotqa_type_embeddings.weight"][tt] * (token_type_ids != 0).unsqueeze(-1)  # :351-352
    e = word_e + pos_e + type_e + loc_e                                # :354
    e = layer_norm(e, sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    return _drop(e, cfg.hidden_dropout_prob, training)


def embed_image(sd, cfg, feat, loc, target, training):
    """BertImageEmbeddings.forward for dataset='plotqa', vilbert.py:1474-1496."""
    pre = "bert.v_embeddings."
    img = linear(sd, pre + "new_image_embeddings", torch.softmax(feat, dim=-1))   # :1476
    e = img + linear(sd, pre + "new_loc_emb", loc) + sd[pre + "color_emb.weight"][target]  # :1478-1486
    e = layer_norm(e, sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    return _drop(e, cfg.hidden_dropout_prob, training)                 # :1470 uses the *text* prob


# ----------------------------------------------------------------------------- encoder layers
def _self_layer(sd, pre, x, add_mask, n_heads, p_att, p_hid, training):
    """BertLayer / BertImageLayer, vilbert.py:361-485 / 488-616."""
    a = pre + "attention."
    ctx = sdpa(linear(sd, a + "self.query", x), linear(sd, a + "self.key", x),
               linear(sd, a + "self.value", x), add_mask, n_heads, p_att, training)
    att = layer_norm(_drop(linear(sd, a + "output.dense", ctx), p_hid, training) + x,
                     sd[a + "output.LayerNorm.weight"], sd[a + "output.LayerNorm.bias"])    # :424-428
    h = gelu_erf(linear(sd, pre + "intermediate.dense", att))                                  # :454-457
    return layer_norm(_drop(linear(sd, pre + "output.dense", h), p_hid, training) + att,
                      sd[pre + "output.LayerNorm.weight"], sd[pre + "output.LayerNorm.bias"])  # :467-471


def _connection_layer(sd, cfg, pre, xv, mask_v, xt, mask_t, training):
    """BertConnectionLayer, vilbert.py:619-788.  Stream 1 = visual, stream 2 = text."""
    b = pre + "biattention."
    nh = cfg.bi_num_attention_heads
    q1, k1, v1 = (linear(sd, b + n + "1", xv) for n in ("query", "key", "value"))    # :662-664
    q2, k2, v2 = (linear(sd, b + n + "2", xt) for n in ("query", "key", "value"))    # :673-675
    ctx1 = sdpa(q2, k1, v1, mask_v, nh, cfg.v_attention_probs_dropout_prob, training)  # text queries, [B,T,Bi] :684-701
    ctx2 = sdpa(q1, k2, v2, mask_t, nh, cfg.attention_probs_dropout_prob, training)    # visual queries, [B,V,Bi] :704-723
    o = pre + "biOutput."
    # cross wiring of vilbert.py:780 -> BertBiOutput.forward(ctx2, xv, ctx1, xt) :746-758
    av = layer_norm(_drop(linear(sd, o + "dense1", ctx2), cfg.v_hidden_dropout_prob, training) + xv,
                    sd[o + "LayerNorm1.weight"], sd[o + "LayerNorm1.bias"])
    at = layer_norm(_drop(linear(sd, o + "dense2", ctx1), cfg.hidden_dropout_prob, training) + xt,
                    sd[o + "LayerNorm2.weight"], sd[o + "LayerNorm2.bias"])
    hv = gelu_erf(linear(sd, pre + "v_intermediate.dense", av))                       # :782
    yv = layer_norm(_drop(linear(sd, pre + "v_output.dense", hv), cfg.v_hidden_dropout_prob, training) + av,
                    sd[pre + "v_output.LayerNorm.weight"], sd[pre + "v_output.LayerNorm.bias"])
    ht = gelu_erf(linear(sd, pre + "t_intermediate.dense", at))                       # :785
    yt = layer_norm(_drop(linear(sd, pre + "t_output.dense", ht), cfg.hidden_dropout_prob, training) + at,
                    sd[pre + "t_output.LayerNorm.weight"], sd[pre + "t_output.LayerNorm.bias"])
    return yv, yt


def encoder_schedule(cfg):
    """Layer execution order of BertEncoder.forward (vilbert.py:852-939) as a list of
    ('t', i) / ('v', i) / ('c', i) steps (with_coattention honoured)."""
    steps, v_start, t_start = [], 0, 0
    for c, (v_end, t_end) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        steps += [("v", i) for i in range(v_start, v_end)]
        steps += [("t", i) for i in range(t_start, t_end)]
        if cfg.with_coattention:
            steps.append(("c", c))
        v_start, t_start = v_end, t_end
    steps += [("v", i) for i in range(v_start, cfg.v_num_hidden_layers)]
    steps += [("t", i) for i in range(t_start, cfg.num_hidden_layers)]
    return steps


def encode(sd, cfg, xt, xv, mask_t, mask_v, training, taps=None):
    for kind, i in encoder_schedule(cfg):
        if kind == "t":
            xt = _self_layer(sd, "bert.encoder.layer.%d." % i, xt, mask_t, cfg.num_attention_heads,
                             cfg.attention_probs_dropout_prob, cfg.hidden_dropout_prob, training)
        elif kind == "v":
            xv = _self_layer(sd, "bert.encoder.v_layer.%d." % i, xv, mask_v, cfg.v_num_attention_heads,
                             cfg.v_attention_probs_dropout_prob, cfg.v_hidden_dropout_prob, training)
        else:
            xv, xt = _connection_layer(sd, cfg, "bert.encoder.c_layer.%d." % i, xv, mask_v, xt, mask_t, training)
        if taps is not None:
            taps["%s%d.t" % (kind, i)] = xt
            taps["%s%d.v" % (kind, i)] = xv
    return xt, xv


# ----------------------------------------------------------------------------- heads + losses
def regressor(sd, hv0, hw0):
    """PlotQA_Regressor_v20.forward, regressor.py:36-42 (LeakyReLU slope 0.01)."""
    def pipe(name, x, last_act):
        for j in (0, 2, 4, 6):
            x = linear(sd, "regressor.%s.%d" % (name, j), x)
            if j != 6:
                x = F.leaky_relu(x, 0.01)
        return last_act(x) if last_act else x
    hw = pipe("txt_pipe", hw0, None)
    hv = pipe("vis_pipe", hv0, None)
    return pipe("fusion", torch.cat((hv, hw), dim=-1), torch.tanh).squeeze(-1)


def heads_and_losses(sd, cfg, params, seq_t, seq_v, R, kind, nsp_label, training, cls_dropout=0.1):
    """Poolers + BertPreTrainingHeads + regression/NSP losses.
    vilbert.py:949-976, 1048-1062, 1583-1657.  All rows are regressed and masked by R[:,1]
    (the reference gathers the needs_regression rows -- identical values and gradients)."""
    pt = torch.relu(linear(sd, "bert.t_pooler.dense", seq_t[:, 0]))
    pv = torch.relu(linear(sd, "bert.v_pooler.dense", seq_v[:, 0]))
    fused = pt * pv if cfg.fusion_method == "mul" else pt + pv
    logits = linear(sd, "cls.bi_seq_relationship", _drop(fused, cls_dropout, training))   # :1045,1060

    needs = R[:, 1] == 1                                             # :1588
    nf = needs.to(R.dtype)
    r = regressor(sd, seq_v[:, 0], seq_t[:, 0])                      # raw CLS / IMG states :1599-1600
    target = R[:, 0] / torch.where(needs, R[:, 3], torch.ones_like(R[:, 3]))   # :1617 (only the needs rows exist there)
    if params["L1"]:
        reg_loss = (r - target).abs()                                # :1526
    else:
        reg_loss = F.smooth_l1_loss(r, target, reduction="none", beta=0.5)   # :1528
    reg_l1 = (r - target).abs()                                      # :1628
    both0 = (r == 0) & (target == 0)
    d5 = reg_l1 / target.abs()                                       # :1632
    d5 = torch.where(target == 0, torch.ones_like(d5), d5)           # :1633
    d5 = torch.where(both0, torch.zeros_like(d5), d5)                # :1634
    ok5 = ((d5 <= 0.05) | both0) & needs                             # :1636
    okt = (reg_l1 <= params["tol_margin"]) & needs                   # :1637
    if kind != "L1":
        reg_loss = torch.where(target.abs() > 1, torch.zeros_like(reg_loss), reg_loss)   # :1639-1641
    zero = torch.zeros_like(r)
    # rows that need no regression carry R = [0, False, 0, 0] in real batches (fig_dataloader.py:631): target = 0 / 0;
    # the reference never computes them (gather / scatter into zeros), so they are selected away, not multiplied away
    reg = [torch.where(needs, r * R[:, 3], zero).detach(),           # :1644 (scatter into zeros)
           torch.where(needs, reg_loss, zero), torch.where(needs, reg_l1, zero).detach(),
           (int(ok5.sum()), int(okt.sum())), torch.where(needs, d5, torch.zeros_like(d5)).detach()]
    nsp = None
    if nsp_label is not None:
        nsp = F.cross_entropy(logits.view(-1, 2), nsp_label.view(-1), ignore_index=-1).unsqueeze(0)   # :1655-1657
    return logits, reg, nsp, r


# ----------------------------------------------------------------------------- step adapter
def text_key_mask(sep_indices, hist_len, T):
    """encoder_decorator.py:57-70,118-120."""
    lengths = torch.gather(sep_indices, 1, hist_len.view(-1, 1)).squeeze(1) + 1
    return torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)


def oracle_step(sd, cfg, params, batch, evaluation=False, training=True, taps=None, cls_dropout=0.1):
    """encoder_decorator.forward (encoder_decorator.py:73-158) + BertForMultiModalPreTraining.forward
    (vilbert.py:1540-1661) on a weight dict whose keys carry no 'bert_pretrained.' prefix.

    Returns the reference's tuple: train -> (loss, lm_loss, nsp_loss, img_loss, nsp_scores,
    regression, legend_loss); eval -> (None, None, None, None, nsp_scores, regression)."""
    tokens, segs = batch["tokens"], batch["segments"]
    T = tokens.shape[1]
    key_t = text_key_mask(batch["sep_indices"], batch["hist_len"], T)
    mask_t = (1.0 - key_t.float())[:, None, None, :] * -10000.0       # vilbert.py:1380-1391
    mask_v = (1.0 - batch["image_mask"].float())[:, None, None, :] * -10000.0   # :1393-1396
    xt = embed_text(sd, cfg, tokens, segs, batch["loc"], training)
    xv = embed_image(sd, cfg, batch["image_feat"], batch["image_loc"], batch["image_target"], training)
    if taps is not None:
        taps["emb.t"], taps["emb.v"] = xt, xv
    xt, xv = encode(sd, cfg, xt, xv, mask_t, mask_v, training, taps)
    kind = "L1" if evaluation else "L1_smooth"                        # encoder_decorator.py:104-106
    labels = None if evaluation else batch["next_sentence_labels"]
    logits, reg, nsp, raw = heads_and_losses(sd, cfg, params, xt, xv, batch["R"], kind, labels,
                                             training, cls_dropout)
    if taps is not None:
        taps["seq_t"], taps["seq_v"], taps["reg_raw"] = xt, xv, raw
    if evaluation:
        return None, None, None, None, logits, reg
    zero11 = torch.zeros(1, 1)
    loss = (params["nsp_loss_coeff"] * nsp + params["reg_loss_coeff"] * reg[1].mean()).sum()   # ed:144-153
    return loss, zero11, nsp, zero11.clone(), logits, reg, torch.zeros(1)


# ----------------------------------------------------------------------------- optimizer / schedule
def lr_factor(step, warmup_steps, t_total):
    """WarmupLinearScheduleNonZero.get_lr factor, utils.py:22-27."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def scheduled_lr(base_lr, step, warmup_steps, t_total, min_lr):
    """utils.py:29 -- floor at min_lr."""
    v = base_lr * lr_factor(step, warmup_steps, t_total)
    return v if v > min_lr else min_lr


NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")   # utils.py:229


def adamw_reference_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (what utils.py:249 constructs), in place, fp32."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p
