#!/usr/bin/env python3
"""Bottom-right causal short chunks on the tiled kernel's SPLIT instances (route 5) against the fp64 oracle, plus
timings.  Run under default and AULE_HIP_FWD_PPSPLIT=0 (plain tiled kernel) in one gpurun call."""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
import torch
import oracle
from aule import _torch as at, _capi

def route(dt, B, Hq, Hkv, Sq, Sk, D, causal=2):
    lib = _capi.get_lib()
    d = _capi.AttnDesc(); d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {torch.float16: 1, torch.bfloat16: 2}[dt]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = causal, -1
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))

def check(dt, B, Hq, Hkv, Sq, Sk, D, scale=None):
    rng = np.random.RandomState(7)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(dt)
    q, k, v = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D)
    sc = 1 / math.sqrt(D) if scale is None else scale
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), "bottom-right", sc)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), "bottom-right", scale)
    o = out.float().cpu().numpy()
    u = 2.0 ** -9 if dt == torch.bfloat16 else 2.0 ** -12
    tol = 1e-3 + u * float(np.abs(v.float().numpy()).max()) + 2 * u * np.abs(ref)
    bad = int((np.abs(o - ref) > tol).sum())
    le = float(np.abs(lse.cpu().numpy() - rl).max())
    print(f"route={route(dt,B,Hq,Hkv,Sq,Sk,D)} {str(dt)[6:]} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} scale={scale}: "
          f"out err {np.abs(o-ref).max():.2e} over_tol={bad} lse err {le:.2e} nan={int(np.isnan(o).sum())}", flush=True)
    return bad == 0 and not np.isnan(o).any() and le < 1e-3

def timed(dt, B, Hq, Hkv, Sq, Sk, D, causal):
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    f = lambda: at.fwd_raw(q, k, v, causal, 1 / math.sqrt(D), want_lse=False)
    for _ in range(5): f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(f"  time B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} causal={causal}: {best*1e3:.1f} us", flush=True)

if __name__ == "__main__":
    bf, fp = torch.bfloat16, torch.float16
    print("PPSPLIT=%s" % os.environ.get("AULE_HIP_FWD_PPSPLIT", "on"))
    ok = True
    if "check" in sys.argv:
        for args in [(bf, 2, 8, 2, 64, 8192, 128), (fp, 1, 8, 2, 17, 3000, 64),      # 68 packed rows: waves span heads
                     (bf, 1, 32, 1, 5, 4100, 128), (fp, 2, 6, 3, 33, 2049, 32), (bf, 1, 4, 4, 200, 5000, 128),
                     (bf, 1, 16, 2, 256, 4096, 64), (bf, 3, 4, 2, 2, 1024, 128), (bf, 8, 32, 8, 1, 8192, 128),   # Sq = 1: non-causal
                     (bf, 1, 8, 8, 64, 1100, 128)]:
            ok &= check(*args)
        ok &= check(bf, 1, 8, 2, 40, 4096, 128, scale=-0.3)
        print("ALL OK" if ok else "FAILURES")
    for args in [(bf, 8, 32, 8, 64, 8192, 128), (bf, 8, 32, 8, 16, 8192, 128), (bf, 1, 32, 8, 64, 8192, 128), (bf, 8, 32, 8, 1, 8192, 128),
                 (bf, 32, 32, 8, 1, 8192, 128), (bf, 4, 32, 8, 128, 16384, 128), (bf, 1, 32, 8, 8, 32768, 128), (bf, 4, 32, 8, 1024, 4096, 128)]:
        timed(*args, "bottom-right")
