#!/usr/bin/env python3
"""Randomised differential test of the whole forward/backward dispatch against the fp64 oracle: random dtype, batch,
GQA ratio, Sq, Sk, head_dim, causal mode, window and scale sign -- the combinations nobody wrote a case for.
Deterministic per seed; prints every failing configuration.   python tools/fuzz_parity.py [n=200] [seed=0]"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oracle
from aule import _torch as at, _capi
from util import BWD_TOL, LSE_TOL, fwd_tol, quantize, torch_dtype

def route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W):
    d = _capi.AttnDesc(); d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = code, W
    return _capi.get_lib().aule_hip_debug_forward_route(ctypes.byref(d))

def draw(rng):
    dtype = rng.choice(["bf16", "bf16", "fp16", "fp32"])
    D = int(rng.choice([32, 64, 128]))
    Hkv = int(rng.choice([1, 1, 2, 3, 4, 8])); g = int(rng.choice([1, 1, 2, 4, 8, 5])); Hq = Hkv * g
    B = int(rng.choice([1, 1, 2, 3, 9]))
    Sq = int(rng.choice([1, 1, 2, 3, 7, 16, 17, 31, 33, 64, 65, 100, 200, 256, 257, 300]))
    Sk = int(rng.choice([1, 5, 63, 64, 65, 130, 500, 1023, 1024, 1025, 2047, 2100, 3000, 4096, 5000, 9000]))
    causal = rng.choice(["none", "none", "top", "br", "br"])
    if causal == "br" and Sk < Sq: Sk = Sq + int(rng.choice([0, 1, 100, 1500, 4000]))
    W = int(rng.choice([-1, -1, -1, 1, 7, 64, 100, 1000]))
    scale = None if rng.rand() < 0.7 else float(rng.choice([0.3, -0.2, 0.05, 1.0]))
    # keep the fp64 judge (fwd + bwd ~ 8 B Hq Sq Sk D flops) within ~0.5 s
    while 8.0 * B * Hq * Sq * Sk * D > 4e8:
        if B > 1: B = 1
        elif Sq > 64: Sq = Sq // 2
        elif Hq > Hkv and g > 1: g = max(1, g // 2); Hq = Hkv * g
        else: Sk = max(Sq if causal == "br" else 1, Sk // 2)
    return dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale

def run(cfg, seed):
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, W, scale = cfg
    cz = {"none": False, "top": True, "br": "bottom-right"}[causal]
    code = {"none": 0, "top": 1, "br": 2}[causal]
    rng = np.random.RandomState(seed)
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), dtype) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))
    tq, tk, tv, tdo = dev(q), dev(k), dev(v), dev(do)
    out, lse = at.fwd_raw(tq, tk, tv, cz, sc, window=W)
    dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, cz, sc, window=W)
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q, k, v, cz, scale, W)
    rq, rk, rv = oracle.bwd_f64(q, k, v, do, cz, scale, W)
    errs = []
    o = out.float().cpu().numpy()
    atol, rtol = fwd_tol(dtype, float(np.abs(v).max()))
    # fp32 arithmetic error grows with the logit magnitude (dO ~ 2^-23 |S| |V| through exp): a plain NumPy fp32
    # softmax(QK^T)V is 2e-7 from the fp64 judge at |S| ~ 5 and 1.5e-5 at |S| ~ 53 on the same inputs, like the kernel.
    # The suite's fixed 1e-5 assumes the default temperature; here the scale is random, so the bound follows |S|.
    g = Hq // Hkv
    smax = max(float(np.abs(q[b, h] @ k[b, h // g].T).max()) for b in range(B) for h in range(Hq)) * abs(sc)
    grow = 1.0
    if dtype == "fp32":
        atol += 2.0 ** -22 * smax * float(np.abs(v).max())
        grow = max(1.0, smax / 8.0)
    if not np.isfinite(o).all(): errs.append("out non-finite")
    elif (np.abs(o - ref) > atol + rtol * np.abs(ref)).any(): errs.append(f"out err {np.abs(o-ref).max():.3e}")
    gl = lse.cpu().numpy(); fin = np.isfinite(rl)
    if not np.all(np.isneginf(gl[~fin])): errs.append("lse of empty rows not -inf")
    if fin.any() and np.abs(gl[fin] - rl[fin]).max() > LSE_TOL[dtype] + 1e-5 * np.abs(rl[fin]).max(): errs.append(f"lse err {np.abs(gl[fin]-rl[fin]).max():.3e}")
    a, r = BWD_TOL[dtype]
    for name, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        gg = got.float().cpu().numpy()
        if not np.isfinite(gg).all(): errs.append(name + " non-finite"); continue
        # dK / dV accumulate g * Sq rows; where the true gradient vanishes (one key: P = 1, dS = dP - delta cancels
        # exactly) what is left is rounding noise ~ eps |dP| |q| sqrt(rows): plain NumPy fp32 leaves 1.3e-5 at 1600 rows
        acc = max(1.0, math.sqrt(g * Sq / 256.0)) if (dtype == "fp32" and name != "dq") else 1.0
        tol = a * grow * acc * max(1.0, float(np.abs(want).max()))
        if (np.abs(gg - want) > tol + r * np.abs(want)).any(): errs.append(f"{name} err {np.abs(gg-want).max():.3e} (tol {tol:.1e})")
    return route(dtype, B, Hq, Hkv, Sq, Sk, D, code, W), errs

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    routes, bad = {}, 0
    for i in range(n):
        cfg = draw(rng)
        try:
            r, errs = run(cfg, 1000 + i)
        except Exception as e:  # noqa: BLE001
            r, errs = -9, [f"EXCEPTION {type(e).__name__}: {str(e)[:160]}"]
        routes[r] = routes.get(r, 0) + 1
        if errs:
            bad += 1
            print(f"FAIL #{i} route={r} cfg={cfg}: {'; '.join(errs)}", flush=True)
    print(f"{n} configurations, {bad} failing; forward routes exercised: {dict(sorted(routes.items()))}")
