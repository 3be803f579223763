import sys, ctypes as C
sys.path.insert(0,'cqa-crct_amd'); sys.path.insert(0,'.')
import torch
from crct import config as CFG, synthetic as S, lib as L
from crct.model import VisualDialogEncoder
from crct.step_adapter import forward as step_forward
dev=torch.device('cuda:0')
cfg=CFG.tiny_config(hidden_dropout_prob=0.1)
params=CFG.default_params(categories=9, device=dev)
m=VisualDialogEncoder(params, config=cfg); core=m.bert_pretrained; core.use_graph=True; core.sync_stats=False
b=S.make_batch(6,9,7,cfg.v_feature_size,categories=9,vocab_size=cfg.vocab_size,seed=3)
lib=L.load(); lib.crct_engine_graph_stats.argtypes=[C.c_void_p,C.POINTER(C.c_int),C.POINTER(C.c_int),C.POINTER(C.c_int)]
for i in range(5):
    out=step_forward(m,b,params); out[0].backward(); torch.cuda.synchronize()
    k,x,br=C.c_int(),C.c_int(),C.c_int()
    lib.crct_engine_graph_stats(core._engine.handle,C.byref(k),C.byref(x),C.byref(br))
    print(i,float(out[0]),k.value,x.value,br.value, lib.crct_last_error())
