import os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
import aule
B,H,S,D = [int(x) for x in sys.argv[1:5]]
q = torch.randn(B, H, S, D, device="cuda", dtype=torch.float16); k = torch.randn_like(q); v = torch.randn_like(q)
for _ in range(50):
    aule.flash_attention(q, k, v, causal=True)
torch.cuda.synchronize()
