#!/usr/bin/env python3
"""One fp32 backward shape a few times (for rocprofv3 passes): python tools/f32_bwd_one.py B Hq Hkv S D causal [reps]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
B, Hq, Hkv, S, D, causal = (int(x) for x in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 6
q = torch.randn(B, Hq, S, D, device="cuda"); k = torch.randn(B, Hkv, S, D, device="cuda"); v = torch.randn_like(k)
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(q, k, v, bool(causal), sc)
do = torch.randn_like(q)
for _ in range(reps):
    at.bwd_raw(q, k, v, out, do, lse, bool(causal), sc)
torch.cuda.synchronize()
