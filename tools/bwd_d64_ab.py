#!/usr/bin/env python3
"""D = 64 backward: timings + a hash of the gradients, for A/B builds that must not change a bit (select the build with AULE_LIBRARY_PATH;
AULE_HIP_BWD_MODE=recompute keeps the dispatch on the recompute pair whatever the size)."""
import hashlib, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
def run(B, Hq, Hkv, Sq, Sk, causal, D=64, dt=torch.bfloat16, time_it=True):
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g); v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    do = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    f = lambda: at.bwd_raw(q, k, v, out, do, lse, causal, sc)
    grads = f(); torch.cuda.synchronize()
    h = hashlib.sha1(b"".join(x.contiguous().view(torch.int16).cpu().numpy().tobytes() for x in grads[:3])).hexdigest()[:16]
    best = float("nan")
    if time_it:
        for _ in range(60): f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D} {str(dt)[6:]} causal={causal}: {best*1e3:8.1f} us  grads {h}", flush=True)
run(4, 32, 8, 2048, 2048, True); run(8, 32, 32, 2048, 2048, True); run(4, 32, 8, 4096, 4096, True, 64, torch.float16); run(2, 16, 16, 4096, 4096, False)
run(2, 8, 2, 1000, 1111, True, time_it=False); run(1, 16, 16, 777, 2048, False, time_it=False); run(1, 32, 8, 2048, 3000, True, time_it=False)
