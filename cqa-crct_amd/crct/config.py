"""Model / run configuration for the CRCT co-attention training step.

Mirrors the reference's configuration surface for the hot path:
  * ``BertConfig``  -- reference ``CRCT/backbone/vilbert.py:127-270`` (every JSON key becomes an
    attribute; keys that are absent fall back to the constructor defaults of ``vilbert.py:131-166``).
  * ``default_params`` -- the subset of ``CRCT/options.py:10-78`` + ``CRCT/config/plotqa.json`` that
    the model / step adapter actually reads (SURVEY.md section 8b "Constructor").

Nothing here touches the GPU.
"""
import copy
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(os.path.dirname(_HERE), "config")

# constructor defaults of the reference BertConfig (vilbert.py:131-166)
_DEFAULTS = dict(
    hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
    hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
    max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
    v_feature_size=1024, v_target_size=1601, v_hidden_size=768, v_num_hidden_layers=3,
    v_num_attention_heads=12, v_intermediate_size=3072, bi_hidden_size=1024,
    bi_num_attention_heads=16, v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu",
    v_hidden_dropout_prob=0.1, v_initializer_range=0.2, v_biattention_id=[0, 1],
    t_biattention_id=[10, 11], predict_feature=False, fast_mode=False, fixed_v_layer=0,
    fixed_t_layer=0, in_batch_pairs=False, fusion_method="mul", intra_gate=False,
    with_coattention=True,
)


class BertConfig(object):
    """Attribute bag with the reference's defaults (vilbert.py:127-258)."""

    def __init__(self, vocab_size_or_config_json_file=-1, **kw):
        for k, v in _DEFAULTS.items():
            setattr(self, k, copy.deepcopy(v))
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as f:
                for k, v in json.load(f).items():
                    setattr(self, k, v)
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
        else:
            # same error convention as vilbert.py:240-243
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")
        for k, v in kw.items():
            setattr(self, k, v)
        self.validate()

    def validate(self):
        # vilbert.py:192-194
        assert len(self.v_biattention_id) == len(self.t_biattention_id)
        if len(self.v_biattention_id):
            assert max(self.v_biattention_id) < self.v_num_hidden_layers
            assert max(self.t_biattention_id) < self.num_hidden_layers
        # vilbert.py:364-368, 491-495, 622-626
        for hs, nh in ((self.hidden_size, self.num_attention_heads),
                       (self.v_hidden_size, self.v_num_attention_heads),
                       (self.bi_hidden_size, self.bi_num_attention_heads)):
            if hs % nh != 0:
                raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                                 "heads (%d)" % (hs, nh))
        if self.fusion_method not in ("mul", "sum"):
            raise AssertionError("fusion_method must be 'mul' or 'sum' (vilbert.py:1052-1057)")
        for flag in ("fast_mode", "in_batch_pairs", "predict_feature", "intra_gate"):
            if getattr(self, flag):
                raise NotImplementedError("%s is off on the CRCT hot path (SURVEY.md 3.2)" % flag)
        if self.fixed_t_layer or self.fixed_v_layer:
            raise NotImplementedError("fixed_*_layer is off on the CRCT hot path (SURVEY.md 3.2)")
        if self.hidden_act != "gelu" or self.v_hidden_act != "gelu":
            raise NotImplementedError("only erf-GELU is built (vilbert.json:3,24)")

    @classmethod
    def from_dict(cls, d):
        c = cls(-1)
        for k, v in d.items():
            setattr(c, k, v)
        c.validate()
        return c

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def __repr__(self):
        return self.to_json_string()


def vilbert_config(**overrides):
    """The shipped model config (values of reference config/vilbert.json) with overrides."""
    c = BertConfig.from_json_file(os.path.join(CONFIG_DIR, "vilbert.json"))
    for k, v in overrides.items():
        setattr(c, k, v)
    c.validate()
    return c


def tiny_config(**overrides):
    """Small config used by the parity fixtures (SURVEY.md 8c 'Tiny config')."""
    d = dict(
        vocab_size=128, plotqa_vocab_types=12, hidden_size=64, num_hidden_layers=3,
        num_attention_heads=4, intermediate_size=128, max_position_embeddings=32,
        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        v_feature_size=32, v_target_size=17, v_hidden_size=96, v_num_hidden_layers=2,
        v_num_attention_heads=4, v_intermediate_size=96, bi_hidden_size=64,
        bi_num_attention_heads=4, v_attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0,
        v_initializer_range=0.02, v_biattention_id=[0, 1], t_biattention_id=[1, 2],
    )
    d.update(overrides)
    return BertConfig.from_dict(d)


def default_params(**overrides):
    """The params-dict keys the model and the step adapter read (options.py / plotqa.json)."""
    p = dict(
        model_config=os.path.join(CONFIG_DIR, "vilbert.json"),
        categories=228, dataset="plotqa", mask_prob_img=0.0, binary_answers=False, qa_file="qa",
        CE_REG=False, L1=True, rank=0, rank_from=0, BOT_MODE=True, max_seq_len=124,
        max_vis_features=44, device="cpu", tol_margin=0.01,
        nsp_loss_coeff=1.0, reg_loss_coeff=1.0, lr=2e-5, image_lr=2e-5, min_lr=1.3e-5, wd=0.01,
        warmup=3000, batch_multiply=1, batch_size=80, world_size=1, ddp=False, seed=0,
    )
    p.update(overrides)
    return p
