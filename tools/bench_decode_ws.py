#!/usr/bin/env python3
"""Decode (split-KV) K+V streaming rate with the working set cycled past the 256 MB Infinity Cache, next to the
single-buffer figure.  A serving loop re-reads one KV cache per step, so the single-buffer rate is what a small
cache sees in production -- but only the cycled figure may be quoted against the HBM roof."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at

def run(B, Hq, Hkv, Sk, D, nbuf, dt=torch.bfloat16):
    q = torch.randn(B, Hq, 1, D, device="cuda", dtype=dt)
    ks = [torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt) for _ in range(nbuf)]
    vs = [torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt) for _ in range(nbuf)]
    sc = 1 / math.sqrt(D)
    iters = max(40 // nbuf, 4) * nbuf
    for i in range(nbuf): at.fwd_raw(q, ks[i], vs[i], False, sc, want_lse=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): at.fwd_raw(q, ks[i % nbuf], vs[i % nbuf], False, sc, want_lse=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    kv_mb = 2 * ks[0].numel() * ks[0].element_size() / 1e6
    print(f"  decode B{B} Hq{Hq} Hkv{Hkv} Sk{Sk} D{D}: K+V {kv_mb:.0f} MB x {nbuf} buffers "
          f"[{'HBM' if kv_mb * nbuf > 600 else 'cache-assisted'}]: {us:.1f} us  {kv_mb/us:.2f} TB/s", flush=True)

if __name__ == "__main__":
    run(8, 32, 8, 8192, 128, 1)
    run(8, 32, 8, 8192, 128, 4)
    run(8, 32, 8, 32768, 128, 1)     # 1.07 GB: past the cache even with one buffer
    run(1, 32, 1, 16384, 64, 1, torch.float16)   # C5b: 4 MB, L2/cache resident by construction
