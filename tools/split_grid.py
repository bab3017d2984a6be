#!/usr/bin/env python3
"""Where does the split-KV path beat the tiled kernel?  Times short-query non-causal shapes under the current
dispatcher; run twice (default, and AULE_HIP_FWD_SPLITKV=0 for the tiled kernel) in the same gpurun call."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at

def t(B, Hq, Hkv, Sq, Sk, D=128, dt=torch.bfloat16):
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    f = lambda: at.fwd_raw(q, k, v, False, sc, want_lse=False)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    g = Hq // Hkv
    print(f"B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} rows/unit={g*Sq} tiled_wgs={B*Hq*((Sq+255)//256)}: {best*1e3:.1f} us", flush=True)

if __name__ == "__main__":
    print("SPLITKV =", os.environ.get("AULE_HIP_FWD_SPLITKV", "(default on)"))
    for B, Hq, Hkv in ((1, 32, 8), (8, 32, 8), (8, 32, 32), (1, 32, 1), (16, 32, 8)):
        for Sq in (1, 8, 16, 32, 64):
            for Sk in (2048, 8192):
                t(B, Hq, Hkv, Sq, Sk)
    t(1, 32, 1, 64, 16384, 64, torch.float16)   # C5c
    t(1, 32, 1, 1, 16384, 64, torch.float16)    # C5b
