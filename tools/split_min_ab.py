#!/usr/bin/env python3
"""Small causal / non-causal grids on the one-wave-per-SIMD forward: how short may a key-range piece be (AULE_HIP_FWD_SPLIT_MIN, read once per process)?
Times the forward of the half-empty shapes of VERDICT r5 item 6 and prints the route and the number of pieces."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
from aule import _capi

def route(B, H, Hkv, S, D, dt, causal):
    d = _capi.AttnDesc(); d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = 2 if dt == torch.bfloat16 else 1
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, H, Hkv, S, S, D
    d.causal, d.window_size = int(causal), -1
    lib = _capi.get_lib()
    r = int(lib.aule_hip_debug_forward_route(ctypes.byref(d)))
    out = (ctypes.c_int32 * 4096)()
    n = int(lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), out, 4096)) if r == 7 else 0
    return r, (out[0] if n > 0 else 1)

def t(f, n=30):
    for _ in range(8): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

print("AULE_HIP_FWD_SPLIT_MIN =", os.environ.get("AULE_HIP_FWD_SPLIT_MIN", "(default 16)"))
for (B, H, Hkv, S, D, dt, causal) in ((1, 32, 32, 2048, 128, torch.float16, True), (1, 16, 16, 2048, 128, torch.float16, True), (1, 8, 8, 4096, 128, torch.float16, True),
                                      (1, 8, 8, 2048, 128, torch.bfloat16, True), (1, 8, 8, 8192, 128, torch.bfloat16, True), (1, 16, 16, 4096, 128, torch.bfloat16, True),
                                      (1, 32, 8, 2048, 128, torch.bfloat16, True), (2, 8, 8, 2048, 128, torch.bfloat16, True), (1, 16, 16, 2048, 64, torch.float16, True),
                                      (1, 8, 8, 4096, 64, torch.float16, True), (1, 8, 8, 2048, 128, torch.bfloat16, False), (1, 16, 16, 2048, 128, torch.bfloat16, False)):
    q = torch.randn(B, H, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    P = S * (S + 1) // 2 if causal else S * S
    us = t(lambda: aule.flash_attention(q, k, v, causal=causal))
    r, n = route(B, H, Hkv, S, D, dt, causal)
    print(f"  B{B} Hq{H} Hkv{Hkv} S{S} D{D} {str(dt)[6:]} causal={int(causal)}: route {r} pieces {n}: {us:7.1f} us = {4.0*B*H*D*P/us/1e6:6.1f} TF", flush=True)
