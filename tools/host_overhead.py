#!/usr/bin/env python3
"""Host-side cost of the autograd step (config C3): is the bench's fwd+bwd leg limited by Python or by the GPU?
   python tools/host_overhead.py      (on the GPU box)
Prints: per-step wall with the queue kept full, the per-step GPU time from events, the host time of one step issued into an
idle GPU (synchronize before each), and the top cumulative host costs (cProfile)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
dev = torch.device("cuda", 0)
B, Hq, Hkv, S, D = 4, 32, 8, 2048, 128
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(B, Hq, S, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
k = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
v = torch.randn(B, Hkv, S, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
do = torch.randn(B, Hq, S, D, device=dev, dtype=torch.bfloat16, generator=g)

def step():
    q.grad = k.grad = v.grad = None
    aule.flash_attention(q, k, v, causal=True).backward(do)

for _ in range(200):
    step()
torch.cuda.synchronize()
N = 100
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(N):
    step()
t_issue = time.perf_counter() - t0
e1.record(); torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"queue kept full: wall {wall / N * 1e6:.0f} us/step, GPU (events) {e0.elapsed_time(e1) / N * 1e3:.0f} us/step, host issue loop {t_issue / N * 1e6:.0f} us/step")
hs = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); hs.append(time.perf_counter() - t0)
hs.sort()
print(f"host time of one step into an idle GPU: median {hs[len(hs) // 2] * 1e6:.0f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# ---- per-call host cost of aule.flash_attention, inference (no autograd node, no LSE) and autograd, on a shape whose kernel is short
# (VERDICT r4 item 8): host issue time per call with the queue kept full, and into an idle GPU
def host_cost(fn, n=2000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return t_issue / n * 1e6, wall / n * 1e6

qs, ks, vs = (torch.randn(1, 8, 128, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
qd = torch.randn(4, 32, 1, 128, device=dev, dtype=torch.bfloat16, generator=g)
kd, vd = (torch.randn(4, 8, 4096, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2))
def inf_small():
    with torch.no_grad():
        aule.flash_attention(qs, ks, vs, causal=True)
def inf_decode():
    with torch.no_grad():
        aule.flash_attention(qd, kd, vd, causal="bottom-right")
from aule import _torch as at
import math
def raw_small():
    at.fwd_raw(qs, ks, vs, 1, 1 / math.sqrt(128), want_lse=False)
qa, ka, va = (x.clone().requires_grad_(True) for x in (qs, ks, vs))
def grad_small():
    aule.flash_attention(qa, ka, va, causal=True)
for name, fn in (("inference B1 H8 S128 (aule.flash_attention)", inf_small), ("inference decode B4 32q/8kv Sk4096 (aule.flash_attention)", inf_decode),
                 ("fwd_raw B1 H8 S128 (binding only)", raw_small), ("autograd forward B1 H8 S128", grad_small)):
    issue, wall = host_cost(fn)
    print(f"host cost per call, {name}: issue loop {issue:.1f} us, wall {wall:.1f} us")
