"""Shared helpers for the parity tests (fixture loading, weight dicts)."""
import json
import os

import numpy as np
import torch

from crct import config as C
from crct import synthetic as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cfg = C.BertConfig.from_dict(meta["cfg"])
    params = dict(meta["params"])
    params["device"] = torch.device("cpu")
    batch = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in.")}
    return z, meta, cfg, params, batch


def param_shapes(cfg, params):
    """{state_dict key (no prefix): shape} for every parameter of the model (SURVEY.md 8b schema)."""
    from crct.layout import parameter_table
    table, _ = parameter_table(cfg, params)
    return {e.name: e.shape for e in table}


def seeded_weights(cfg, params, base_seed=7, requires_grad=True):
    sd = {}
    for k, shp in param_shapes(cfg, params).items():
        t = S.seeded_tensor(k, shp, base_seed)
        sd[k] = t.requires_grad_(requires_grad)
    return sd
