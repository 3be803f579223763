"""CPU: parameter schema / flat layout, config surface, and the C ABI (library loads, every symbol the
header declares is exported, ctypes mirrors have the header's sizes).  No GPU compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from crct import config as CFG
from crct import layout as LY
from crct import lib as L
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parameter_table_matches_reference_schema():
    z = np.load(os.path.join(GOLDEN, "tiny_L1.npz"))
    ref_names = [k[2:] for k in z.files if k.startswith("w.")]          # reference named_parameters() order
    table, total = LY.parameter_table(CFG.tiny_config(), CFG.default_params(categories=9))
    assert [e.name for e in table] == ref_names
    for e in table:
        assert tuple(z["w." + e.name].shape) == e.shape
    # never-used tensors = the ones whose gradient is None in the reference
    unused = sorted(k[9:] for k in z.files if k.startswith("gradnorm.") and float(z[k]) < 0)
    assert sorted(e.name for e in table if not e.used) == unused


def test_full_layout_properties():
    cfg, params = CFG.vilbert_config(), CFG.default_params()
    table, total = LY.parameter_table(cfg, params)
    assert len(table) == 560 and sum(e.numel for e in table) == 252666750
    assert sum(1 for e in table if e.used) == 524 and sum(e.numel for e in table if e.used) == 238317315
    assert sum(1 for e in table if e.language) == 201
    by_off = sorted(table, key=lambda e: e.offset)
    end = 0
    for e in by_off:
        assert e.offset >= end
        end = e.offset + e.numel
    assert end <= total
    off = {e.name: e for e in table}
    for i in range(12):                                                  # fused QKV: q, k, v adjacent
        q, k, v = (off["bert.encoder.layer.%d.attention.self.%s.weight" % (i, n)] for n in ("query", "key", "value"))
        assert k.offset == q.offset + q.numel and v.offset == k.offset + k.numel
    c = "bert.encoder.c_layer.3.biattention."
    assert off[c + "key2.bias"].offset == off[c + "query2.bias"].offset + 1024
    lo, hi = LY.used_span(table)
    assert all(e.offset >= hi for e in table if not e.used)             # unused tensors at the tail: never all-reduced
    # first-use order: heads after every encoder layer, embeddings first
    assert off["regressor.fusion.6.weight"].offset > off["bert.encoder.layer.11.output.dense.weight"].offset
    assert off["bert.embeddings.word_embeddings.weight"].offset == 0
    sched = LY.encoder_schedule(cfg)
    assert sched[:7] == [("t", 0), ("t", 1), ("t", 2), ("t", 3), ("t", 4), ("t", 5), ("c", 0)] and sched[-2:] == [("v", 5), ("t", 11)]


def test_config_errors_match_reference_conventions():
    with pytest.raises(ValueError):
        CFG.tiny_config(hidden_size=66)                 # not a multiple of the heads (vilbert.py:364-368)
    with pytest.raises(AssertionError):
        CFG.tiny_config(v_biattention_id=[0, 5])        # vilbert.py:192-194
    with pytest.raises(ValueError):
        CFG.BertConfig(3.5)
    c = CFG.BertConfig.from_json_file(os.path.join(CFG.CONFIG_DIR, "vilbert.json"))
    assert c.fusion_method == "mul" and c.bi_num_attention_heads == 32 and c.v_feature_size == 1024


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "crct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crct_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "libcrct_hip.so must be built (python -c 'import __graft_entry__ as g; g.build()')"
    lib = C.CDLL(L.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "symbol %s declared in include/crct_hip.h is not exported" % name
    assert set(L.PROTOTYPES) <= set(declared)
    handle = L.load()
    assert handle.crct_abi_version() == 7


def test_shipped_library_carries_no_lab_hook():
    """The timing ablations (leave-one-out launches, GEMM kernels without DMA / epilogue, the sleeping AdamW stand-in) exist only in the
    -DCRCT_GEMM_LAB build under tools/lab/ (`make -C cqa-crct_amd/csrc lab`): the package's library must not be able to skip work, and it
    reads no CRCT_* environment variable."""
    with open(L.LIB_PATH, "rb") as f:
        blob = f.read()
    for needle in (b"crct_lab_skip", b"CRCT_LAB_", b"CRCT_GEMM_DBG", b"lab_spin_kernel"):
        assert needle not in blob, needle


def test_struct_mirrors_and_error_path():
    # sizes from the header's field lists (LP64)
    assert C.sizeof(L.GemmArgs) == (7 * 8 + 5 * 8 + 10 * 4 + 4 + 4 + 4 + 4 + 8 + 8 + 8 + 5 * 8 + 8 + 4 + 4 + 8 + 8 + 4 + 4)     # ... seed, rowsum_out, fp8 (+pad), 5 pointers, ld_q, site, split_k, 2 pointers, addend_f32, c_cached
    assert C.sizeof(L.Batch) == 10 * 8 + 3 * 4 + 4 + 3 * 8 + 4 + 4      # ... B, T, V, pad, sep_indices, hist_len, image_mask, sep_stride, image_feat_bf16
    assert C.sizeof(L.ModelDims) == 16 * 4 + 64 * 4 + 2 * 4 + 5 * 4
    assert C.sizeof(L.LnFwdArgs) == 6 * 8 + 2 * 4 + 4 * 4 + 8 + 3 * 8 + 8 + 8       # ... x_f32 (+pad), y_f32
    assert C.sizeof(L.LnBwdArgs) == 8 * 8 + 2 * 4 + 6 * 4 + 8 + 3 * 8 + 8            # ... x_f32 (+pad)
    assert C.sizeof(L.AttnQuant) == 13 * 8
    lib = L.load()
    g = L.GemmArgs()
    g.M, g.N, g.K = 8, 6, 8            # N % 4 != 0 -> rejected before anything touches a GPU
    g.A = g.B = g.C = 16
    g.lda = g.ldb = g.ldc = 8
    assert lib.crct_gemm_bf16(C.byref(g), None) != 0
    assert b"multiple of 4" in lib.crct_last_error()
    assert lib.crct_layernorm_bwd_blocks(1600) == 256 and lib.crct_colsum_blocks(80) == 10
    assert lib.crct_gemm_pick_tile(1600, 3072) in (0, 1, 2, 3)


def test_model_refuses_cpu():
    from crct.model import VisualDialogEncoder
    with pytest.raises(RuntimeError):
        VisualDialogEncoder(CFG.default_params(categories=9, device="cpu"), config=CFG.tiny_config())


def test_adamw_plan_on_host():
    from crct import ops
    seg, off = ops.adamw_plan([4096 * 2 + 5, 10, 4096])
    assert seg.tolist() == [0, 0, 0, 1, 2] and off.tolist() == [0, 4096, 8192, 0, 0]


def test_synthetic_batch_schema():
    from crct import synthetic as S
    b = S.make_batch(5, 20, 36, 64, seed=1)
    assert b["tokens"].shape == (5, 20) and b["image_feat"].shape == (5, 36, 64) and b["R"].shape == (5, 4)
    assert b["sep_indices"].shape == (5, 50) and b["hist_len"].shape == (5, 1) and b["image_target"][0, 0] == 228
    assert set(torch.unique(b["segments"]).tolist()) <= {-1, 0, 1, 4, 7, 8, 9, 10, 11}
    assert (b["segments"][:, 0] == 0).all() and (b["tokens"][:, 0] == 101).all()
    b2 = S.make_batch(5, 20, 36, 64, seed=1)
    assert all(torch.equal(b[k], b2[k]) for k in b)


def test_checkpoint_schema_fixture_matches_the_parameter_table():
    """tests/golden/ckpt_schema.json (structure of a checkpoint written by the reference's train.py path) lists exactly
    the tensors of the flat layout, in state_dict order, plus the tied LM decoder key; AdamW state exists only for the
    tensors that receive gradients (train.py:284-291, utils.py:228-249)."""
    import json
    schema = json.load(open(os.path.join(GOLDEN, "ckpt_schema.json")))
    cfg = CFG.tiny_config()
    params = CFG.default_params(categories=9)
    table, _ = LY.parameter_table(cfg, params)
    by_name = {"bert_pretrained." + e.name: e for e in table}
    keys = [k for k, _, _ in schema["model_state_dict"]]
    tied = "bert_pretrained.cls.predictions.decoder.weight"
    assert tied in keys and set(keys) - {tied} == set(by_name)
    for k, shape, dtype in schema["model_state_dict"]:
        if k != tied:
            assert tuple(shape) == tuple(by_name[k].shape) and dtype == "torch.float32"
    # one optimizer group per named parameter (the tied key is the same Parameter), state only where gradients flow
    named = [k for k in keys if k != tied]
    assert len(schema["optimizer_param_groups"]) == len(named)
    assert [g[2] for g in schema["optimizer_param_groups"]] == [[i] for i in range(len(named))]
    # named_parameters() order == state_dict order without the tied key
    used_ids = [i for i, k in enumerate(named) if by_name[k].used]
    assert schema["optimizer_state_ids"] == used_ids
