#!/usr/bin/env python3
"""Very long sequences: forward at S = 131072 (sampled rows against the fp64 judge: O and LSE), causal and not, and the
backward at S = 65536 against the judge's dQ rows / dK, dV column samples via linearity-free direct evaluation on a
sub-problem (the causal prefix: the first n rows/keys of a causal problem are a causal problem of their own)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
import torch
import oracle
from aule import _torch as at

def fwd_check(B, Hq, Hkv, S, D, causal, dt=torch.bfloat16, nrows=24):
    g = torch.Generator(device="cuda").manual_seed(S + int(bool(causal)))
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt, generator=g)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, generator=g); v = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, generator=g)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    rng = np.random.RandomState(1)
    rows = np.unique(np.concatenate([[0, S - 1, (B * Hq - 1) * S + S - 1, (B * Hq - 1) * S], rng.randint(0, B * Hq * S, nrows)])).astype(np.int64)
    o_r, l_r = oracle.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), rows, causal)
    go = out.float().cpu().numpy().reshape(-1, D)[rows]; gl = lse.cpu().numpy().reshape(-1)[rows]
    u = 2.0 ** -9
    tol = 1e-3 + u * float(v.float().abs().max()) + 2 * u * np.abs(o_r)
    bad = int((np.abs(go - o_r) > tol).sum())
    fl = 4.0 * B * Hq * S * S * D * (0.5 if causal else 1.0)
    print(f"fwd B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={causal}: {ms:.2f} ms = {fl/ms/1e9:.0f} TFLOP/s; {len(rows)} sampled rows: "
          f"out err {np.abs(go-o_r).max():.2e} over_tol={bad}, lse err {np.abs(gl-l_r).max():.2e}, finite={bool(np.isfinite(go).all())}", flush=True)
    return bad == 0 and np.abs(gl - l_r).max() < 2e-3

def bwd_check(B, Hq, Hkv, S, D, n=384, dt=torch.bfloat16):
    """causal: gradients restricted to the first n queries/keys with dO = 0 beyond them equal the gradients of the
    n-row causal problem -- checked against the fp64 judge on that prefix; the full-size call exercises the long loops."""
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt, generator=g)
    k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, generator=g); v = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt, generator=g)
    do = torch.zeros(B, Hq, S, D, device="cuda", dtype=dt)
    do[:, :, :n] = torch.randn(B, Hq, n, D, device="cuda", dtype=dt, generator=g)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, True, sc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dq, dk, dv = at.bwd_raw(q, k, v, out, do, lse, True, sc)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    qn, kn, vn, don = (x[:, :, :n].float().cpu().numpy() for x in (q, k, v, do))
    rq, rk, rv = oracle.bwd_f64(qn, kn, vn, don, True)
    ok = True
    for name, got, want in (("dq", dq[:, :, :n], rq), ("dk", dk[:, :, :n], rk), ("dv", dv[:, :, :n], rv)):
        e = float(np.abs(got.float().cpu().numpy() - want).max()); tol = 2e-2 * max(1.0, float(np.abs(want).max()))
        print(f"bwd S{S} prefix {n}: {name} err {e:.2e} (tol {tol:.1e})", flush=True); ok &= e <= tol
    tail = max(float(dq[:, :, n:].float().abs().max()), float(dk[:, :, n:].float().abs().max()), float(dv[:, :, n:].float().abs().max()))
    print(f"bwd B{B} Hq{Hq} Hkv{Hkv} S{S}: {ms:.1f} ms; gradients beyond the prefix (must be 0): {tail:.1e}", flush=True)
    return ok and tail == 0.0

if __name__ == "__main__":
    ok = fwd_check(1, 4, 2, 131072, 128, True)
    ok &= fwd_check(1, 4, 2, 131072, 128, False)
    ok &= fwd_check(1, 8, 1, 65536, 64, True, torch.float16)
    ok &= bwd_check(1, 4, 2, 65536, 128)
    print("ALL OK" if ok else "PROBLEMS")
