#!/usr/bin/env python3
"""Sliding-window forward / forward+backward timings on the shapes of the reference's README (python/README.md:36-40: S 2K / 4K / 8K,
window 256).  Round 6: the window instances of the one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip WIN) by default; AULE_HIP_W4_WINDOW=0 = the
ping-pong kernel's route (fa_fwd_pp_gfx950.hip), the A/B partner.
TFLOP/s counts the visible scores only (a band of `window` keys per query)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule


def t(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, S, D, W) in ((4, 32, 2048, 128, 256), (4, 32, 4096, 128, 256), (4, 32, 8192, 128, 256), (4, 32, 8192, 128, 1024), (4, 32, 4096, 64, 256)):
    q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    vis = sum(min(i + 1, W) for i in range(S))          # visible scores per (batch, head)
    fl = 4.0 * B * H * D * vis
    us = t(lambda: aule.flash_attention(q, k, v, causal=True, window_size=W))
    us_full = t(lambda: aule.flash_attention(q, k, v, causal=True))
    qg, kg, vg = (x.clone().requires_grad_(True) for x in (q, k, v))
    do = torch.randn_like(q)
    def fb():
        o = aule.flash_attention(qg, kg, vg, causal=True, window_size=W)
        o.backward(do)
        qg.grad = kg.grad = vg.grad = None
    us_fb = t(fb, 10)
    print(f"  bf16 B{B} H{H} S{S} D{D} causal window {W}: fwd {us:8.1f} us = {fl/us/1e6:6.1f} TF of visible scores ({us_full/us:4.1f}x faster than the full causal call, {us_full:7.1f} us); "
          f"fwd+bwd {us_fb:8.1f} us = {3.5*fl/us_fb/1e6:6.1f} TF", flush=True)
