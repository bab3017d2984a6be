import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch, aule
B, Hq, Hkv, D, bs = 8, 32, 8, 128, 16
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = ctx // bs
kc = torch.randn(B * nb, bs, Hkv, D, device="cuda", dtype=torch.float16); vc = torch.randn_like(kc)
q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16)
bt = torch.randperm(B * nb, device="cuda").to(torch.int32).view(B, nb)
cl = torch.full((B,), ctx, device="cuda", dtype=torch.int32)
for _ in range(30): aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
torch.cuda.synchronize()
