"""CPU: the parts of the drop-in boundary that need no GPU.

* BERT-base initialisation (crct/pretrained.py) with the rules of the reference's ``from_pretrained`` (CRCT/backbone/vilbert.py:1154-1285:
  archive resolution, gamma / beta rename :1219-1231, ``bert.`` prefix rule :1259-1263, name-and-shape matching, size mismatch =
  RuntimeError :1277-1283) on a synthetic BERT-shaped state dict (real key names and shapes of bert-base-uncased, random values).
* ``FlatGradDDP`` as the ``nn.Module`` wrapper CRCT/train.py:138-143,173,289 and CRCT/evaluation.py:56-61 use in
  DistributedDataParallel's place: ``.module``, call-through, train / eval, state_dict prefix, ``no_sync``, save / resume through
  ``.module.state_dict()`` -- over a stand-in core (the real one needs the GPU: tests/test_binding_gpu.py replays the loop there).
"""
import collections
import os
import tarfile

import pytest
import torch
import torch.distributed as dist

from crct import config as C
from crct import pretrained as P
from helpers import param_shapes


def bert_base_state_dict(legacy_names=True, seed=0):
    """Keys and shapes of bert-base-uncased's ``pytorch_model.bin`` (BertForPreTraining: 12 layers, H = 768, vocab 30522, 512
    positions, 2 token types), LayerNorm parameters under their TensorFlow-era names gamma / beta as in the original archive."""
    g = torch.Generator().manual_seed(seed)
    ln_w, ln_b = ("gamma", "beta") if legacy_names else ("weight", "bias")
    H, I, sd = 768, 3072, collections.OrderedDict()

    def add(key, *shape):
        sd[key] = torch.randn(*shape, generator=g) * 0.02

    add("bert.embeddings.word_embeddings.weight", 30522, H)
    add("bert.embeddings.position_embeddings.weight", 512, H)
    add("bert.embeddings.token_type_embeddings.weight", 2, H)
    add("bert.embeddings.LayerNorm." + ln_w, H)
    add("bert.embeddings.LayerNorm." + ln_b, H)
    for i in range(12):
        p = "bert.encoder.layer.%d." % i
        for name in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            add(p + name + ".weight", H, H)
            add(p + name + ".bias", H)
        add(p + "attention.output.LayerNorm." + ln_w, H)
        add(p + "attention.output.LayerNorm." + ln_b, H)
        add(p + "intermediate.dense.weight", I, H)
        add(p + "intermediate.dense.bias", I)
        add(p + "output.dense.weight", H, I)
        add(p + "output.dense.bias", H)
        add(p + "output.LayerNorm." + ln_w, H)
        add(p + "output.LayerNorm." + ln_b, H)
    add("bert.pooler.dense.weight", H, H)
    add("bert.pooler.dense.bias", H)
    add("cls.predictions.bias", 30522)
    add("cls.predictions.transform.dense.weight", H, H)
    add("cls.predictions.transform.dense.bias", H)
    add("cls.predictions.transform.LayerNorm." + ln_w, H)
    add("cls.predictions.transform.LayerNorm." + ln_b, H)
    sd["cls.predictions.decoder.weight"] = sd["bert.embeddings.word_embeddings.weight"]
    add("cls.seq_relationship.weight", 2, H)
    add("cls.seq_relationship.bias", 2)
    return sd


def model_shapes():
    cfg, params = C.vilbert_config(), C.default_params()
    shapes = collections.OrderedDict(param_shapes(cfg, params))
    shapes["cls.predictions.decoder.weight"] = shapes["bert.embeddings.word_embeddings.weight"]      # the tied second key (vilbert.py:1029)
    return shapes


def test_bert_base_checkpoint_maps_onto_the_text_stream_and_the_lm_head():
    shapes = model_shapes()
    assert len(shapes) == 561                                    # 561 state_dict entries / 560 parameters
    sd = bert_base_state_dict()
    copies, missing, unexpected, errors = P.plan_load(shapes, sd)
    assert not errors
    got = dict(copies)
    # 4 embedding tensors + 12 layers x 16 + the LM head's 6 (tied decoder included) = 202
    assert len(copies) == 4 + 12 * 16 + 6
    assert set(k for k in got if k.startswith("bert.encoder.layer.")) == set(k for k in shapes if k.startswith("bert.encoder.layer."))
    for k in ("bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight", "bert.embeddings.LayerNorm.weight",
              "bert.embeddings.LayerNorm.bias", "cls.predictions.bias", "cls.predictions.transform.LayerNorm.weight", "cls.predictions.decoder.weight"):
        assert k in got, k
    # the renamed LayerNorm tensors carry the gamma / beta VALUES
    assert torch.equal(got["bert.encoder.layer.7.output.LayerNorm.weight"], sd["bert.encoder.layer.7.output.LayerNorm.gamma"])
    assert torch.equal(got["bert.embeddings.LayerNorm.bias"], sd["bert.embeddings.LayerNorm.beta"])
    # what BERT has and CRCT has not / the other way round
    assert sorted(unexpected) == sorted(["bert.embeddings.token_type_embeddings.weight", "bert.pooler.dense.weight", "bert.pooler.dense.bias",
                                         "cls.seq_relationship.weight", "cls.seq_relationship.bias"])
    assert len(missing) == 561 - len(copies)
    assert all(k.startswith(("bert.encoder.v_layer.", "bert.encoder.c_layer.", "bert.v_embeddings.", "bert.t_pooler.", "bert.v_pooler.", "regressor.",
                             "cls.bi_seq_relationship.", "cls.imagePredictions.", "bert.embeddings.txt_location_embeddings.",
                             "bert.embeddings.plotqa_type_embeddings.")) for k in missing), [k for k in missing][:5]
    # modern names load the same tensors
    copies2, _, _, _ = P.plan_load(shapes, bert_base_state_dict(legacy_names=False))
    assert [k for k, _ in copies2] == [k for k, _ in copies]


def test_prefix_rule_size_mismatch_and_archive_resolution(tmp_path):
    shapes = model_shapes()
    sd = bert_base_state_dict()
    # vilbert.py:1259-1263: a model WITHOUT a `bert` attribute (its keys start below it) reads a checkpoint whose keys start with "bert."
    inner = collections.OrderedDict((k[len("bert."):], v) for k, v in shapes.items() if k.startswith("bert."))
    copies, missing, unexpected, errors = P.plan_load(inner, sd, model_has_bert=False)
    assert not errors and len(copies) == 4 + 12 * 16 and "encoder.layer.0.attention.self.query.weight" in dict(copies)
    assert all(k.startswith("bert.") for k in unexpected) and "cls.predictions.bias" not in unexpected        # only keys under the prefix are reported
    # a model WITH `bert` never gets the prefix
    assert P.plan_load(shapes, sd, model_has_bert=True)[0][0][0].startswith("bert.")
    # size mismatch: reported like torch's _load_from_state_dict and raised by the loader (vilbert.py:1277-1283)
    bad = collections.OrderedDict(sd)
    bad["bert.embeddings.position_embeddings.weight"] = torch.zeros(1024, 768)
    _, _, _, errors = P.plan_load(shapes, bad)
    assert len(errors) == 1 and "size mismatch for bert.embeddings.position_embeddings.weight" in errors[0] and "(1024, 768)" in errors[0]
    # archive resolution (vilbert.py:1183-1200): directory, .bin file, .tar.gz archive; a missing path is an error, not None
    small = collections.OrderedDict((k, v) for k, v in sd.items() if "layer" not in k and "word" not in k and "decoder" not in k and k != "cls.predictions.bias")
    d = tmp_path / "bert-base-uncased"
    d.mkdir()
    torch.save(small, str(d / P.WEIGHTS_NAME))
    for source in (str(d), str(d / P.WEIGHTS_NAME)):
        back = P.read_state_dict(source)
        assert list(back) == list(small) and torch.equal(back["bert.pooler.dense.bias"], small["bert.pooler.dense.bias"])
    tgz = tmp_path / "bert-base-uncased.tar.gz"
    with tarfile.open(str(tgz), "w:gz") as tf:
        tf.add(str(d / P.WEIGHTS_NAME), arcname=P.WEIGHTS_NAME)
    assert list(P.read_state_dict(str(tgz))) == list(small)
    with pytest.raises(FileNotFoundError, match="not found"):
        P.read_state_dict(str(tmp_path / "nowhere"))
    # an nn.Module is accepted like a state dict (vilbert.py:1214-1215)
    lin = torch.nn.Linear(3, 2)
    assert list(P.read_state_dict(lin)) == ["weight", "bias"]


class _StandInCore(torch.nn.Module):
    """What FlatGradDDP touches of a CrctModel at construction: the flat parameter buffer (broadcast from rank 0) and the shadow flag."""

    def __init__(self):
        super().__init__()
        self._flat_p = torch.arange(12, dtype=torch.float32)
        self.w = torch.nn.Parameter(self._flat_p[:8].view(2, 4))
        self.b = torch.nn.Parameter(self._flat_p[8:])
        self._ddp = None
        self.invalidated = 0

    @property
    def flat_params(self):
        return self._flat_p

    def _invalidate_shadow(self):
        self.invalidated += 1


class _StandInEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.bert_pretrained = _StandInCore()

    def forward(self, x, scale=1.0, **kw):
        return (x * scale, self.training, sorted(kw))


def test_flat_grad_ddp_is_a_module_wrapper_with_the_surface_the_reference_loops_use(tmp_path):
    from crct.ddp import FlatGradDDP
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = _StandInEncoder()
        ddp = FlatGradDDP(model, device_ids=None, find_unused_parameters=True)             # train.py:139-142 with the import swapped
        assert isinstance(ddp, torch.nn.Module) and ddp.module is model                       # train.py:289 `crct_model.module`
        assert model.bert_pretrained._ddp is ddp and model.bert_pretrained.invalidated == 1   # attached to the core; parameters broadcast from rank 0
        out = ddp(torch.ones(2), scale=3.0, image_target=None)                               # train.py:173 -> encoder_decorator.py:125-142: keyword call
        assert torch.equal(out[0], torch.full((2,), 3.0)) and out[1] is True and out[2] == ["image_target"]
        ddp.eval()                                                                            # evaluation.py: dialog_encoder.eval()
        assert model.training is False and ddp(torch.ones(1))[1] is False
        ddp.train()
        assert model.training is True
        # every tensor once; the wrapper's own keys carry DistributedDataParallel's prefix, `.module.state_dict()` does not
        assert [k for k, _ in ddp.named_parameters()] == ["module.bert_pretrained.w", "module.bert_pretrained.b"]
        assert list(ddp.state_dict()) == ["module.bert_pretrained.w", "module.bert_pretrained.b"]
        assert list(ddp.module.state_dict()) == ["bert_pretrained.w", "bert_pretrained.b"]
        assert ddp.require_sync is True
        with ddp.no_sync():
            assert ddp.require_sync is False
        assert ddp.require_sync is True
        # save through .module.state_dict() (train.py:287-291), resume into a fresh model (train.py:91-103)
        path = str(tmp_path / "plotqa_encoder_0_1.ckpt")
        torch.save({"model_state_dict": ddp.module.state_dict(), "iter_id": 1}, path)
        fresh = _StandInEncoder()
        with torch.no_grad():
            fresh.bert_pretrained.w.zero_()
        pre = torch.load(path, map_location="cpu")["model_state_dict"]
        model_dict = fresh.state_dict()
        pre = {k: v for k, v in pre.items() if k in model_dict}
        assert len(pre) == 2
        model_dict.update(pre)
        fresh.load_state_dict(model_dict)
        assert torch.equal(fresh.bert_pretrained.w, model.bert_pretrained.w)
        # what is not this package's model is refused; a device list of another size too
        with pytest.raises(TypeError, match="VisualDialogEncoder"):
            FlatGradDDP(torch.nn.Linear(2, 2))
        with pytest.raises(ValueError, match="exactly one device"):
            FlatGradDDP(_StandInEncoder(), device_ids=[0, 1])
    finally:
        dist.destroy_process_group()
