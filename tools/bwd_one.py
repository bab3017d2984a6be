#!/usr/bin/env python3
"""One backward shape in a loop (for rocprofv3 counter passes):  python tools/bwd_one.py B Hq Hkv S causal [D] [n]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
B, Hq, Hkv, S, causal = (int(x) for x in sys.argv[1:6])
D = int(sys.argv[6]) if len(sys.argv) > 6 else 128
n = int(sys.argv[7]) if len(sys.argv) > 7 else 40
q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(q, k, v, causal, sc)
for _ in range(n):
    at.bwd_raw(q, k, v, out, do, lse, causal, sc)
torch.cuda.synchronize()
