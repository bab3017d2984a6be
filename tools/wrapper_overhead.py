#!/usr/bin/env python3
"""CPU cost per call of the public wrapper on a launch-bound shape (C5b): wall time of a loop through
aule.flash_attention against a loop over the raw binding, both under no_grad.  Run with and without
AULE_HIP_ALWAYS_AUTOGRAD=1 in one gpurun call."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
from aule import _torch as at

def loop(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

print("ALWAYS_AUTOGRAD=%s" % os.environ.get("AULE_HIP_ALWAYS_AUTOGRAD", "0"))
for name, (B, Hq, Hkv, Sq, Sk, D, dt) in {"C5b": (1, 32, 1, 1, 16384, 64, torch.float16), "C5c": (1, 32, 1, 64, 16384, 64, torch.float16),
                                          "B8 decode": (8, 32, 8, 1, 8192, 128, torch.bfloat16)}.items():
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    with torch.no_grad():
        pub = loop(lambda: aule.flash_attention(q, k, v, causal=False))
        raw = loop(lambda: at.fwd_raw(q, k, v, False, sc, want_lse=False))
    print(f"  {name}: aule.flash_attention {pub:.1f} us/call, raw binding {raw:.1f} us/call", flush=True)
