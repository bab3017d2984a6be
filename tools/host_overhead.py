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
