#!/usr/bin/env python3
"""Do the forward paths survive hipGraph capture (torch.cuda.graph), in particular the two-launch short-query paths
with their stream-ordered workspace?  Captures N decode steps, replays, compares with eager, times a replayed step."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
from aule import _torch as at

def run(name, B, Hq, Hkv, Sq, Sk, D, dt, causal, steps=20):
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    eager, _ = at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    torch.cuda.synchronize()
    outs = []
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                       # warm-up on the side stream, as torch's capture rules ask
        for _ in range(3): at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            for _ in range(steps):
                o, _ = at.fwd_raw(q, k, v, causal, sc, want_lse=False)
                outs.append(o)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: CAPTURE FAILED: {type(e).__name__}: {str(e)[:200]}", flush=True)
        return False
    for o in outs: o.zero_()
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(o, eager) for o in outs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us_graph = e0.elapsed_time(e1) / (10 * steps) * 1e3
    e0.record()
    for _ in range(10 * steps): at.fwd_raw(q, k, v, causal, sc, want_lse=False)
    e1.record(); torch.cuda.synchronize()
    us_eager = e0.elapsed_time(e1) / (10 * steps) * 1e3
    print(f"{name}: capture ok, replay == eager: {same}; per step: graph {us_graph:.1f} us, eager loop {us_eager:.1f} us", flush=True)
    return same

if __name__ == "__main__":
    ok = True
    ok &= run("C5b  (route 5)", 1, 32, 1, 1, 16384, 64, torch.float16, False)
    ok &= run("C5c  (route 5)", 1, 32, 1, 64, 16384, 64, torch.float16, False)
    ok &= run("B8 decode Sk8192 (route 4)", 8, 32, 8, 1, 8192, 128, torch.bfloat16, False)
    ok &= run("B8 Sq64 bottom-right (route 5 causal)", 8, 32, 8, 64, 8192, 128, torch.bfloat16, "bottom-right")
    ok &= run("B1 H32 S2048 causal (plain)", 1, 32, 32, 2048, 2048, 128, torch.bfloat16, True, steps=5)
    print("ALL OK" if ok else "PROBLEMS")
