#!/usr/bin/env python3
"""A few short-query shapes, 30 calls each, for rocprofv3 --kernel-trace --stats (per-kernel split of the
two-launch paths: SPLIT/wave kernel vs fa_fwd_splitkv_combine)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
which = sys.argv[1]
shapes = {"c5b": (1, 32, 1, 1, 16384, 64, torch.float16), "c5c": (1, 32, 1, 64, 16384, 64, torch.float16),
          "b8s64": (8, 32, 8, 64, 8192, 128, torch.bfloat16), "b32dec": (32, 32, 8, 1, 8192, 128, torch.bfloat16),
          "b1h8s1024": (1, 8, 8, 1024, 32768, 128, torch.bfloat16)}
B, Hq, Hkv, Sq, Sk, D, dt = shapes[which]
q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
for _ in range(30): at.fwd_raw(q, k, v, False, 1 / math.sqrt(D), want_lse=False)
torch.cuda.synchronize()
