#!/usr/bin/env python3
"""A/B timing of one library build (select it with AULE_LIBRARY_PATH): the headline forward shapes and the C3
backward, HIP-event timed.  Run both builds in the same gpurun call: boxes differ by a few percent."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at

def timed(fn, iters):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best

def fwd(B, Hq, Hkv, Sq, Sk, causal, D=128, window=-1, dt=torch.bfloat16):
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    ms = timed(lambda: at.fwd_raw(q, k, v, causal, sc, want_lse=False, window=window), 20)
    print(f"  fwd B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} causal={causal} W={window}: {ms*1e3:.1f} us", flush=True)

def bwd(B, Hq, Hkv, S, causal, D=128):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    ms = timed(lambda: at.bwd_raw(q, k, v, out, do, lse, causal, sc), 20)
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} S{S} causal={causal}: {ms*1e3:.1f} us", flush=True)

if __name__ == "__main__":
    print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
    fwd(4, 32, 32, 4096, 4096, True)
    fwd(4, 32, 32, 4096, 4096, False)
    fwd(4, 32, 8, 2048, 2048, True)
    fwd(1, 32, 8, 8192, 8192, True)
    fwd(4, 32, 32, 4096, 4096, True, window=512)
    bwd(4, 32, 8, 2048, True)
    bwd(2, 16, 16, 4096, False)
    for extra in sys.argv[1:]:
        if extra == "br":
            fwd(4, 32, 8, 1024, 4096, "bottom-right")
            fwd(8, 32, 8, 64, 8192, "bottom-right")
            fwd(8, 32, 8, 64, 8192, False)
