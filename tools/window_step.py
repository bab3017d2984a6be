#!/usr/bin/env python3
"""One sliding-window training step repeated (B4 H32 S8192 D128 bf16 causal, window from argv, default 256): the command behind the per-kernel
window profiles (rocprofv3 --kernel-trace --stats wraps it)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aule-attention_amd"))
import aule
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
g = torch.Generator(device="cuda").manual_seed(5)
q, k, v = (torch.randn(4, 32, S, 128, device="cuda", dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(3))
d = torch.randn(4, 32, S, 128, device="cuda", dtype=torch.bfloat16, generator=g)
for _ in range(30):
    q.grad = k.grad = v.grad = None
    aule.flash_attention(q, k, v, causal=True, window_size=W).backward(d)
torch.cuda.synchronize()
