import sys, os, math, torch
sys.path.insert(0, "aule-attention_amd")
from aule import _torch as at
B,Hq,Hkv,S,D=4,32,8,2048,128
q=torch.randn(B,Hq,S,D,device="cuda",dtype=torch.bfloat16); k=torch.randn(B,Hkv,S,D,device="cuda",dtype=torch.bfloat16); v=torch.randn_like(k); do=torch.randn_like(q)
out,lse=at.fwd_raw(q,k,v,True,1/math.sqrt(D))
for _ in range(5): at.bwd_raw(q,k,v,out,do,lse,True,1/math.sqrt(D))
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): at.bwd_raw(q,k,v,out,do,lse,True,1/math.sqrt(D))
e1.record(); torch.cuda.synchronize()
print("  bwd C3: %.1f us" % (e0.elapsed_time(e1)/30*1e3))
