#!/usr/bin/env python3
"""Quick GPU triage (not a test): max error of every kernel family vs the oracle and a first
timing of the headline shapes.  Never raises; prints one line per case so that a single
gpurun call tells which kernels are wrong."""
import math
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np
import torch
import oracle
import aule
from aule import _torch as at

DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def one(dtype, B, Hq, Hkv, Sq, Sk, D, causal, bwd=True):
    tag = f"{dtype} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} causal={int(causal)}"
    try:
        rng = np.random.RandomState(1)
        mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(DT[dtype])
        q, k, v, do = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D), mk(B, Hq, Sq, D)
        qn, kn, vn, dn = (x.float().numpy() for x in (q, k, v, do))
        sc = 1 / math.sqrt(D)
        qc, kc, vc, dc = (x.cuda() for x in (q, k, v, do))
        out, lse = at.fwd_raw(qc, kc, vc, causal, sc)
        torch.cuda.synchronize()
        ref, rl = oracle.fwd_f64(qn, kn, vn, causal)
        o = out.float().cpu().numpy()
        msg = f"fwd err {np.abs(o - ref).max():.3e} lse err {np.abs(lse.cpu().numpy() - rl).max():.3e} nan={int(np.isnan(o).sum())}"
        if bwd:
            dq, dk, dv = at.bwd_raw(qc, kc, vc, out, dc, lse, causal, sc)
            torch.cuda.synchronize()
            rq, rk, rv = oracle.bwd_f64(qn, kn, vn, dn, causal)
            msg += " | dq %.3e dk %.3e dv %.3e (ref max %.2f %.2f %.2f)" % (
                np.abs(dq.float().cpu().numpy() - rq).max(), np.abs(dk.float().cpu().numpy() - rk).max(),
                np.abs(dv.float().cpu().numpy() - rv).max(), np.abs(rq).max(), np.abs(rk).max(), np.abs(rv).max())
        print(tag, "::", msg, flush=True)
    except Exception as e:  # noqa
        print(tag, ":: EXCEPTION", repr(e), flush=True)
        traceback.print_exc()


def timeit(name, fn, flops, iters=10):
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"TIME {name}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
    except Exception as e:  # noqa
        print(f"TIME {name}: EXCEPTION {e!r}", flush=True)


def main():
    print(torch.cuda.get_device_name(0), aule.get_backend_info(), flush=True)
    for dtype in ("bf16", "fp16", "fp32"):
        for D in (128, 64, 32):
            one(dtype, 1, 2, 2, 64, 64, D, False)
            one(dtype, 1, 4, 2, 300, 300, D, True)
    one("bf16", 2, 8, 2, 320, 333, 128, True)
    one("bf16", 1, 4, 1, 130, 70, 128, True)
    one("bf16", 1, 2, 2, 1024, 1024, 128, True)

    def cflops(B, H, S, D):
        return 4.0 * B * H * D * (S * (S + 1) // 2)

    B, H, S, D = 4, 32, 4096, 128
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    timeit("C2 fwd bf16 B4 H32 S4096 D128 causal", lambda: at.fwd_raw(q, k, v, True, 1 / math.sqrt(D)), cflops(B, H, S, D))
    timeit("C2-shape fwd non-causal", lambda: at.fwd_raw(q, k, v, False, 1 / math.sqrt(D)), 4.0 * B * H * D * S * S)
    B, Hq, Hkv, S = 4, 32, 8, 2048
    q, do = (torch.randn(B, Hq, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    k, v = (torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    out, lse = at.fwd_raw(q, k, v, True, 1 / math.sqrt(D))
    timeit("C3 fwd GQA 32/8 S2048", lambda: at.fwd_raw(q, k, v, True, 1 / math.sqrt(D)), cflops(B, Hq, S, D))
    timeit("C3 bwd GQA 32/8 S2048", lambda: at.bwd_raw(q, k, v, out, do, lse, True, 1 / math.sqrt(D)), 2.5 * cflops(B, Hq, S, D))
    B, Hq, Hkv, S, D = 1, 32, 1, 16384, 64
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=torch.float16)
    k, v = (torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.float16) for _ in range(2))
    timeit("C5 fwd MQA 32/1 S16384 D64 fp16 non-causal", lambda: at.fwd_raw(q, k, v, False, 1 / math.sqrt(D)), 4.0 * B * Hq * D * S * S, iters=5)
    x = torch.randn(1, 8, 2048, 64, device="cuda")
    timeit("fp32 fwd B1 H8 S2048 D64 causal", lambda: at.fwd_raw(x, x, x, True, 0.125), cflops(1, 8, 2048, 64), iters=5)


if __name__ == "__main__":
    main()
