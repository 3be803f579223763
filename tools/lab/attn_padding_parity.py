"""Kernel-level attention parity under the padding real PlotQA batches carry (CRCT/utils.py:152-160 pads every sample to 124 tokens) and
with near-uniform attention (scores scaled by 0.01: what name-keyed seeded weights produce): forward and the three gradients of
attention_long.hip (T = 124 and, forced, T = 112) and attention_mfma.hip (T = 112) against float64 PyTorch on the same bf16 operands --
cosine and norm ratio per output.  Output of round 6: profiles/r6_attn_padding_parity.txt."""
import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "cqa-crct_amd"), ROOT): sys.path.insert(0, p)
from crct import ops, lib as L
lib = L.load()
def ref_attn(q,k,v,km,heads,d):
    B,Tq,_=q.shape; Tk=k.shape[1]
    qh=q.view(B,Tq,heads,d).permute(0,2,1,3); kh=k.view(B,Tk,heads,d).permute(0,2,1,3); vh=v.view(B,Tk,heads,d).permute(0,2,1,3)
    s=qh@kh.transpose(-1,-2)/math.sqrt(d)+(1.0-km.double())[:,None,None,:]*-10000.0
    return (torch.softmax(s,-1)@vh).permute(0,2,1,3).reshape(B,Tq,heads*d)
def cos(a,b):
    a,b=a.double().flatten(),b.double().flatten(); return float(a@b/(a.norm()*b.norm()))
for T, lens in ((124,[124,71,96,110]),(124,[124]*4),(112,[112,64,87,99]),(112,[112]*4)):
  for scale in (1.0, 0.1):
    B,heads,d=4,16,48; Hh=heads*d
    g=torch.Generator().manual_seed(5)
    q=(torch.randn(B,T,Hh,generator=g)*scale).cuda().bfloat16(); k=(torch.randn(B,T,Hh,generator=g)*scale).cuda().bfloat16(); v=torch.randn(B,T,Hh,generator=g).cuda().bfloat16()
    do=torch.randn(B,T,Hh,generator=g).cuda().bfloat16()
    km=torch.zeros(B,T,dtype=torch.uint8,device="cuda")
    for b in range(B):
        km[b,:lens[b]]=1; do[b,lens[b]:]=0
    qr,kr,vr=(t.double().clone().requires_grad_(True) for t in (q,k,v))
    r=ref_attn(qr,kr,vr,km,heads,d); r.backward(do.double())
    paths=[("long",1)] + ([("short",0)] if T<=112 else [])
    for name,force in paths:
        lib.crct_attention_force_long(force)
        ctx=ops.attention_fwd(q,k,v,km,heads,d)
        dq,dk,dv=ops.attention_bwd(q,k,v,km,do,heads,d)
        print("T=%d lens=%s scale=%.1f %-5s ctx cos %.6f | dq cos %.5f norm %.4f | dk cos %.5f norm %.4f | dv cos %.5f norm %.4f" % (T,lens,scale,name,cos(ctx,r),
              cos(dq,qr.grad),float(dq.double().norm()/qr.grad.norm()),cos(dk,kr.grad),float(dk.double().norm()/kr.grad.norm()),cos(dv,vr.grad),float(dv.double().norm()/vr.grad.norm())))
    lib.crct_attention_force_long(0)
