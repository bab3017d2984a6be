#!/usr/bin/env python3
"""Window instances of the one-wave-per-SIMD forward: where the embedded-request tail starts to pay (AULE_HIP_W4_WTAIL = whole tiles a wave needs; read once
per process: run once per setting).  Times the forward at a row of window lengths."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule

def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

print("AULE_HIP_W4_WTAIL =", os.environ.get("AULE_HIP_W4_WTAIL", "(default)"))
for (B, H, S, D, dt) in ((4, 32, 8192, 128, torch.bfloat16), (4, 32, 4096, 64, torch.float16)):
    q = torch.randn(B, H, S, D, device="cuda", dtype=dt); k = torch.randn_like(q); v = torch.randn_like(q)
    for W in (256, 320, 384, 512, 640, 768, 1024, 2048):
        vis = sum(min(i + 1, W) for i in range(S))
        us = t(lambda: aule.flash_attention(q, k, v, causal=True, window_size=W))
        print(f"  B{B} H{H} S{S} D{D} W{W}: {us:8.1f} us = {4.0*B*H*D*vis/us/1e6:6.1f} TF of visible scores", flush=True)
