#!/usr/bin/env python3
"""Packed-rows + KV-split mode of the tiled kernel (route 5) against the fp64 oracle.  Run with
AULE_HIP_FWD_SPLITKV=0 so that short-query shapes reach it instead of the wave-per-chunk kernel."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
import torch
import oracle
from aule import _torch as at, _capi

def route(dt, B, Hq, Hkv, Sq, Sk, D):
    lib = _capi.get_lib()
    d = _capi.AttnDesc(); d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {torch.float16: 1, torch.bfloat16: 2}[dt]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = 0, -1
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))

def check(dt, B, Hq, Hkv, Sq, Sk, D, scale=None, mag=1.0):
    rng = np.random.RandomState(5)
    mk = lambda *s: torch.from_numpy((rng.randn(*s) * mag).astype(np.float32)).to(dt)
    q, k, v = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D)
    sc = 1 / math.sqrt(D) if scale is None else scale
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), False, sc)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), False, scale)
    o = out.float().cpu().numpy()
    u = 2.0 ** -9 if dt == torch.bfloat16 else 2.0 ** -12
    tol = 1e-3 + u * float(np.abs(v.float().numpy()).max()) + 2 * u * np.abs(ref)
    bad = int((np.abs(o - ref) > tol).sum())
    print(f"route={route(dt,B,Hq,Hkv,Sq,Sk,D)} {str(dt)[6:]} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} scale={scale} mag={mag}: "
          f"out err {np.abs(o-ref).max():.2e} over_tol={bad} lse err {np.abs(lse.cpu().numpy()-rl).max():.2e} "
          f"nan={int(np.isnan(o).sum())}", flush=True)
    return bad == 0 and not np.isnan(o).any() and np.abs(lse.cpu().numpy() - rl).max() < 1e-3

if __name__ == "__main__":
    bf, fp = torch.bfloat16, torch.float16
    ok = True
    for args in [(bf, 8, 32, 8, 64, 8192, 128), (bf, 8, 32, 8, 32, 2048, 128), (bf, 2, 8, 2, 64, 5000, 128),   # ragged Sk
                 (fp, 1, 32, 1, 64, 16384, 64), (bf, 1, 12, 4, 50, 3001, 64), (fp, 3, 6, 3, 17, 2049, 32),
                 (bf, 1, 16, 2, 9, 1024, 64), (bf, 8, 32, 32, 64, 8192, 128), (bf, 1, 8, 8, 300, 9000, 128),   # Sq > 64, MHA
                 (bf, 1, 4, 1, 700, 4100, 128), (fp, 2, 4, 2, 129, 2000, 128)]:
        ok &= check(*args)
    ok &= check(bf, 2, 8, 2, 40, 4096, 128, scale=-0.2)
    ok &= check(bf, 1, 8, 2, 64, 4096, 128, mag=6.0)      # large logits: the fixed-reference pass must redo
    print("ALL OK" if ok else "FAILURES")
