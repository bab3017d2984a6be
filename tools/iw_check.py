#!/usr/bin/env python3
"""Triage of the in-wave ping-pong forward kernel (AULE_HIP_FWD_KERNEL=iw): errors vs the fp64 oracle on
shapes that exercise every tile kind (single tile, ragged Sk, causal pairs, GQA), a large-logit case that
must take the SAFE path, then timings of the headline shapes against the ping-pong kernel."""
import math, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
import torch
import oracle
from aule import _torch as at

def check(B, Hq, Hkv, Sq, Sk, causal, mag=1.0, D=128):
    rng = np.random.RandomState(3)
    mk = lambda *s: torch.from_numpy((rng.randn(*s) * mag).astype(np.float32)).to(torch.bfloat16)
    q, k, v = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), causal, sc)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal)
    o = out.float().cpu().numpy()
    print(f"B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} causal={int(causal)} mag={mag}: fwd err {np.abs(o-ref).max():.3e} "
          f"lse err {np.abs(lse.cpu().numpy()-rl).max():.3e} nan={int(np.isnan(o).sum())}", flush=True)

def bench(B, H, S, causal, iters=20):
    q = torch.randn(B, H, S, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q); v = torch.randn_like(q)
    sc = 1 / math.sqrt(128)
    for _ in range(5): at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * H * S * S * 128 * (0.5 if causal else 1.0)
    print(f"  bench B{B} H{H} S{S} causal={int(causal)}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s", flush=True)

if __name__ == "__main__":
    print("kernel:", os.environ.get("AULE_HIP_FWD_KERNEL", "(default)"))
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        bench(4, 32, 4096, True, 50); bench(4, 32, 4096, False, 30); bench(2, 32, 8192, True, 20)
        sys.exit(0)
    check(1, 2, 2, 64, 64, False)
    check(1, 2, 2, 64, 64, True)
    check(1, 2, 2, 256, 256, True)
    check(1, 2, 2, 300, 300, True)
    check(1, 4, 2, 512, 512, True)
    check(2, 4, 1, 1024, 1024, True)
    check(1, 2, 2, 200, 333, False)
    check(1, 2, 2, 777, 130, False)
    check(1, 2, 2, 1024, 1024, False)
    check(1, 2, 2, 512, 512, True, mag=6.0)
    check(1, 2, 2, 512, 512, False, mag=12.0)
    bench(4, 32, 4096, True, 30); bench(4, 32, 4096, False, 20)
