"""Developer stress for the stream-mode bit-equality screen: one fresh process = one baseline + N multi-stream runs."""
import sys, os
sys.path.insert(0, "cqa-crct_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from crct import config as C, synthetic as S, lib as L
from crct.step_adapter import forward as step_forward
from test_step_gpu import build_model
vis, wg = int(sys.argv[1]), int(sys.argv[2])
cfg = C.vilbert_config(v_feature_size=2048)
params = C.default_params()
model, params = build_model(cfg, params, weights=None, seed=3)
core = model.bert_pretrained
core.cls_dropout = 0.1
batch = S.make_batch(16, 20, 36, 2048, seed=5)
lib = L.load()
atomic = ("word_embeddings", "position_embeddings", "plotqa_type_embeddings", "color_emb")
def run(v, w):
    core.zero_flat_grads(); core._calls = 0
    loss = step_forward(model, batch, params)[0]
    lib.crct_engine_set_streams(core._engine.handle, v, w)
    loss.backward(); torch.cuda.synchronize()
    return float(loss.detach()), core.flat_grads.clone()
def diff(a, b):
    out = []
    for e in core.table:
        if not e.used or any(x in e.name for x in atomic): continue
        x, y = a[e.offset:e.offset+e.numel], b[e.offset:e.offset+e.numel]
        if not torch.equal(x, y):
            d = (x - y).abs(); out.append((e.name, float(d.max()), int((d > 0).sum())))
    return out
step_forward(model, batch, params)
lib.crct_engine_set_streams(core._engine.handle, 0, 0)
l0, g0 = run(0, 0)
prev = None
for rep in range(8):
    lib.crct_engine_set_streams(core._engine.handle, vis, wg)
    l1, g1 = run(vis, wg)
    d = diff(g0, g1)
    if d:
        print("mode", (vis, wg), "rep", rep, "differs from single-stream baseline:", d[:4], "| equals previous multi run:", prev is not None and not diff(prev, g1))
    prev = g1
lib.crct_engine_set_streams(core._engine.handle, 0, 0)
l2, g2 = run(0, 0)
print("mode", (vis, wg), "second single-stream baseline equals first:", not diff(g0, g2))
