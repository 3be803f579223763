"""Parameter schema and flat HBM layout of the CRCT model.

``parameter_table(cfg, params)`` lists every parameter of the reference model
``BertForMultiModalPreTraining`` (keys / shapes / ``named_parameters()`` order: SURVEY.md 8b; module
definitions CRCT/backbone/vilbert.py:297-316, 361-485, 488-616, 619-788, 949-1077, 1333-1346,
1444-1471, 1502-1537 and CRCT/backbone/regressor.py:5-34) together with the element offset of the
tensor inside ONE flat buffer.

Layout rules (MI355X-first):
  * tensors are ordered by first use in the forward pass (embeddings, then the encoder schedule
    of vilbert.py:852-939, then poolers / heads / regressor), so gradients complete back-to-front
    during backward and contiguous ranges can be all-reduced while backward is still running;
  * query / key / value weights (and biases) of one attention are adjacent: the fused-QKV GEMM reads
    them as one [3H, H] operand, while ``state_dict`` still exposes the reference's separate keys;
  * every tensor starts on a 64-element boundary (256 B fp32 / 128 B bf16);
  * parameters that never receive a gradient on this path (biOutput.q_dense1/2,
    cls.predictions.*, cls.imagePredictions.*, v_embeddings.type_embeddings) sit at the end.
"""
import math
from collections import namedtuple

Entry = namedtuple("Entry", "name shape offset numel used decay language")

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")      # CRCT/utils.py:229
ALIGN = 64


def encoder_schedule(cfg):
    """('t'|'v'|'c', index) steps in execution order (vilbert.py:852-939)."""
    steps, vs, ts = [], 0, 0
    for c, (ve, te) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        steps += [("v", i) for i in range(vs, ve)]
        steps += [("t", i) for i in range(ts, te)]
        if cfg.with_coattention:
            steps.append(("c", c))
        vs, ts = ve, te
    steps += [("v", i) for i in range(vs, cfg.v_num_hidden_layers)]
    steps += [("t", i) for i in range(ts, cfg.num_hidden_layers)]
    return steps


def is_language_weight(name):
    """Membership rule of the reference's config/language_weights.json (BERT-base tensors get
    params['lr'], everything else params['image_lr']; utils.py:231-241).  The JSON lists the text
    embeddings' word/position/LayerNorm tensors, every ``bert.encoder.layer.*`` tensor and the
    ``cls.predictions`` bias/transform tensors."""
    if name in ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                "bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias"):
        return True
    if name.startswith("bert.encoder.layer."):
        return True
    if name.startswith("cls.predictions.") and not name.endswith("decoder.weight"):
        return True
    return False


def _linear(prefix, out_f, in_f):
    return [(prefix + ".weight", (out_f, in_f)), (prefix + ".bias", (out_f,))]


def _ln(prefix, h):
    return [(prefix + ".weight", (h,)), (prefix + ".bias", (h,))]


def _self_layer(p, H, I):
    a = p + "attention."
    return (_linear(a + "self.query", H, H) + _linear(a + "self.key", H, H) + _linear(a + "self.value", H, H)
            + _linear(a + "output.dense", H, H) + _ln(a + "output.LayerNorm", H)
            + _linear(p + "intermediate.dense", I, H) + _linear(p + "output.dense", H, I) + _ln(p + "output.LayerNorm", H))


def _conn_layer(p, cfg):
    H, Hv, Hb, I, Iv = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size, cfg.intermediate_size, cfg.v_intermediate_size
    b, o = p + "biattention.", p + "biOutput."
    return (_linear(b + "query1", Hb, Hv) + _linear(b + "key1", Hb, Hv) + _linear(b + "value1", Hb, Hv)
            + _linear(b + "query2", Hb, H) + _linear(b + "key2", Hb, H) + _linear(b + "value2", Hb, H)
            + _linear(o + "dense1", Hv, Hb) + _ln(o + "LayerNorm1", Hv) + _linear(o + "q_dense1", Hv, Hb)
            + _linear(o + "dense2", H, Hb) + _ln(o + "LayerNorm2", H) + _linear(o + "q_dense2", H, Hb)
            + _linear(p + "v_intermediate.dense", Iv, Hv) + _linear(p + "v_output.dense", Hv, Iv) + _ln(p + "v_output.LayerNorm", Hv)
            + _linear(p + "t_intermediate.dense", I, H) + _linear(p + "t_output.dense", H, I) + _ln(p + "t_output.LayerNorm", H))


def _pipe(prefix, widths):
    out = []
    for j in range(4):
        out += _linear("%s.%d" % (prefix, 2 * j), widths[j + 1], widths[j])
    return out


def registration_order(cfg, params):
    """[(name, shape)] in the reference's ``named_parameters()`` order (tied decoder weight omitted,
    as ``named_parameters`` does)."""
    H, Hv, Hb = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size
    e, v = "bert.embeddings.", "bert.v_embeddings."
    out = [(e + "word_embeddings.weight", (cfg.vocab_size, H)),
           (e + "position_embeddings.weight", (cfg.max_position_embeddings, H))]
    out += _linear(e + "txt_location_embeddings", H, 4)
    out += [(e + "plotqa_type_embeddings.weight", (cfg.plotqa_vocab_types, H))] + _ln(e + "LayerNorm", H)
    out += _linear(v + "new_image_embeddings", Hv, cfg.v_feature_size)
    out += [(v + "type_embeddings.weight", (13, Hv)), (v + "color_emb.weight", (params["categories"] + 1, Hv))]
    out += _linear(v + "new_loc_emb", Hv, 4) + _ln(v + "LayerNorm", Hv)
    for i in range(cfg.num_hidden_layers):
        out += _self_layer("bert.encoder.layer.%d." % i, H, cfg.intermediate_size)
    for i in range(cfg.v_num_hidden_layers):
        out += _self_layer("bert.encoder.v_layer.%d." % i, Hv, cfg.v_intermediate_size)
    for i in range(len(cfg.v_biattention_id)):
        out += _conn_layer("bert.encoder.c_layer.%d." % i, cfg)
    out += _linear("bert.t_pooler.dense", Hb, H) + _linear("bert.v_pooler.dense", Hb, Hv)
    out += [("cls.predictions.bias", (cfg.vocab_size,))]
    out += _linear("cls.predictions.transform.dense", H, H) + _ln("cls.predictions.transform.LayerNorm", H)
    out += _linear("cls.bi_seq_relationship", 2, Hb)
    out += _linear("cls.imagePredictions.transform.dense", Hv, Hv) + _ln("cls.imagePredictions.transform.LayerNorm", Hv)
    out += _linear("cls.imagePredictions.decoder", cfg.v_target_size, Hv)
    out += _pipe("regressor.txt_pipe", (H, H, 512, 256, 256))
    out += _pipe("regressor.vis_pipe", (Hv, Hv, 512, 256, 256))
    out += _pipe("regressor.fusion", (512, 512, 256, 256, 1))
    return out


def is_unused(name):
    """Tensors that never receive a gradient on the CRCT path."""
    return (".biOutput.q_dense" in name or name.startswith("cls.predictions.") or name.startswith("cls.imagePredictions.")
            or name == "bert.v_embeddings.type_embeddings.weight")


def _fused_order(names):
    """Reorder a layer's tensors so q/k/v weights, then q/k/v biases are adjacent."""
    def key(n):
        for grp, tags in (("A", ("query", "key", "value")),):
            for t_i, t in enumerate(tags):
                for suffix in ("", "1", "2"):
                    if (".%s%s.weight" % (t, suffix)) in n:
                        return (0, suffix, 0, t_i)
                    if (".%s%s.bias" % (t, suffix)) in n:
                        return (0, suffix, 1, t_i)
        return (1, "", 0, 0)
    qkv = sorted([n for n in names if key(n)[0] == 0], key=key)
    rest = [n for n in names if key(n)[0] == 1]
    return qkv + rest


def parameter_table(cfg, params):
    reg = registration_order(cfg, params)
    shapes = dict(reg)
    names = [n for n, _ in reg]

    def with_prefix(p):
        return [n for n in names if n.startswith(p)]

    flat = []
    flat += with_prefix("bert.embeddings.")
    flat += [n for n in with_prefix("bert.v_embeddings.") if not is_unused(n)]
    for kind, i in encoder_schedule(cfg):
        pre = {"t": "bert.encoder.layer.%d.", "v": "bert.encoder.v_layer.%d.", "c": "bert.encoder.c_layer.%d."}[kind] % i
        flat += _fused_order([n for n in with_prefix(pre) if not is_unused(n)])
    flat += with_prefix("bert.t_pooler.") + with_prefix("bert.v_pooler.") + with_prefix("cls.bi_seq_relationship.")
    flat += with_prefix("regressor.")
    seen = set(flat)
    # layers outside the schedule (e.g. with_coattention=False) and the never-used tensors go last
    flat += [n for n in names if n not in seen]

    offsets, top = {}, 0
    prev = None
    for n in flat:
        numel = int(math.prod(shapes[n]))
        # keep fused q/k/v groups gap-free: only align the first member of a group
        fused_follow = prev is not None and _is_qkv_follow(prev, n)
        if not fused_follow:
            top = (top + ALIGN - 1) // ALIGN * ALIGN
        offsets[n] = top
        top += numel
        prev = n
    total = (top + ALIGN - 1) // ALIGN * ALIGN
    table = [Entry(n, tuple(shapes[n]), offsets[n], int(math.prod(shapes[n])), not is_unused(n),
                   not any(nd in n for nd in NO_DECAY), is_language_weight(n)) for n in names]
    return table, total


def _is_qkv_follow(prev, cur):
    for a, b in (("query", "key"), ("key", "value")):
        for suffix in ("", "1", "2"):
            for kind in ("weight", "bias"):
                if prev.endswith(".%s%s.%s" % (a, suffix, kind)) and cur.endswith(".%s%s.%s" % (b, suffix, kind)) \
                        and prev.rsplit(".", 2)[0] == cur.rsplit(".", 2)[0]:
                    return True
    return False


def used_span(table):
    """[lo, hi) element range that holds every gradient-receiving tensor."""
    used = [e for e in table if e.used]
    return min(e.offset for e in used), max(e.offset + e.numel for e in used)
