#!/usr/bin/env python3
"""Backward timings of grouped-head shapes (the ones the one-wave-per-SIMD dK/dV kernel takes): run once with the default
dispatch and once with AULE_HIP_BWD_DKV=old in the same session for a same-box A/B."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
print("AULE_HIP_BWD_DKV =", os.environ.get("AULE_HIP_BWD_DKV", "(default)"))
def bwd(B, Hq, Hkv, S, causal, dt=torch.bfloat16):
    D = 128
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    f = lambda: at.bwd_raw(q, k, v, out, do, lse, causal, sc)
    for _ in range(60): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} {str(dt)[6:]} causal={causal}: {best*1e3:.1f} us", flush=True)
bwd(4, 32, 8, 2048, True); bwd(4, 32, 8, 4096, True); bwd(8, 32, 4, 2048, False); bwd(1, 32, 1, 8192, True, torch.float16); bwd(2, 64, 8, 8192, True)
if os.environ.get("AULE_HIP_BWD_DKV") == "new":
    bwd(4, 32, 32, 4096, True); bwd(2, 16, 16, 4096, False)
