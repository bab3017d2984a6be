#!/usr/bin/env python3
"""Triage of the split-KV short-query forward: errors vs the fp64 oracle, then timings with and without it."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np, torch, oracle
from aule import _torch as at
DT = {"fp16": torch.float16, "bf16": torch.bfloat16}
for dt, B, Hq, Hkv, Sq, Sk, D in [("fp16", 1, 32, 1, 1, 16384, 64), ("fp16", 1, 32, 1, 64, 16384, 64), ("bf16", 2, 8, 2, 1, 4096, 128),
                                   ("bf16", 1, 4, 4, 3, 1500, 128), ("fp16", 2, 6, 3, 17, 2049, 32), ("bf16", 1, 16, 2, 9, 1024, 64),
                                   ("bf16", 3, 2, 2, 64, 5000, 128)]:
    rng = np.random.RandomState(9)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(DT[dt])
    q, k, v = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D)
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), False, 1 / math.sqrt(D))
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), False)
    o = out.float().cpu().numpy()
    print(f"{dt} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D}: out err {np.abs(o-ref).max():.2e} lse err {np.abs(lse.cpu().numpy()-rl).max():.2e} nan={int(np.isnan(o).sum())}", flush=True)
for (B, Hq, Hkv, Sq, Sk, D, dt) in [(1, 32, 1, 1, 16384, 64, torch.float16), (1, 32, 1, 64, 16384, 64, torch.float16), (8, 32, 8, 1, 8192, 128, torch.bfloat16)]:
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    for _ in range(5): at.fwd_raw(q, k, v, False, 0.125, want_lse=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): at.fwd_raw(q, k, v, False, 0.125, want_lse=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    byt = 2 * (2 * B * Hq * Sq * D + 2 * B * Hkv * Sk * D)
    print(f"  B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D}: {us:.1f} us/call  {byt/us/1e3:.0f} GB/s algorithmic", flush=True)
