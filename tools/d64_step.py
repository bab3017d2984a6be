#!/usr/bin/env python3
"""The D = 64 training step bench.py times as `extra.d64_*` (B8 H32 S2048 D64 bf16 causal, fwd + bwd through the public API), 60 times:
the command behind profiles/r*_fwdbwd_d64_* (rocprofv3 wraps it; tools/sessions/*)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aule-attention_amd"))
import aule
g = torch.Generator(device="cuda").manual_seed(99)
q, k, v = (torch.randn(8, 32, 2048, 64, device="cuda", dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(3))
d = torch.randn(8, 32, 2048, 64, device="cuda", dtype=torch.bfloat16, generator=g)
for _ in range(60):
    q.grad = k.grad = v.grad = None
    aule.flash_attention(q, k, v, causal=True).backward(d)
torch.cuda.synchronize()
