#!/usr/bin/env python3
"""Backward timings for an A/B of the backward kernels (select the build with AULE_LIBRARY_PATH)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
def bwd(B, Hq, Hkv, S, causal, D=128, dt=torch.bfloat16):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    f = lambda: at.bwd_raw(q, k, v, out, do, lse, causal, sc)
    for _ in range(60): f()   # long warm-up: the chip's clock controller needs ~40 ms of load to settle
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} {str(dt)[6:]} causal={causal}: {best*1e3:.1f} us", flush=True)
bwd(4, 32, 8, 2048, True); bwd(4, 32, 32, 4096, True); bwd(2, 16, 16, 4096, False); bwd(4, 32, 8, 2048, True, 64); bwd(1, 32, 1, 8192, True, 128, torch.float16)
