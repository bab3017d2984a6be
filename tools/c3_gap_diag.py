#!/usr/bin/env python3
"""Why does bench.py's C3 fwd+bwd leg read ~670 us per step when the same autograd step in a plain loop takes ~604 (tools/host_overhead.py,
same box, same session)?  Times the step (a) with events at the ends only, (b) with an event recorded after every step (what bench.py's timed()
does for its per-launch statistics), (c) again after the allocator has held and released C2-sized tensors.
    python tools/c3_gap_diag.py      (on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
dev = torch.device("cuda", 0)
B, Hq, Hkv, S, D = 4, 32, 8, 2048, 128
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda h, s=S, b=B: torch.randn(b, h, s, D, device=dev, dtype=torch.bfloat16, generator=g)
q, k, v = mk(Hq).requires_grad_(True), mk(Hkv).requires_grad_(True), mk(Hkv).requires_grad_(True)
do = mk(Hq)

def step():
    q.grad = k.grad = v.grad = None
    aule.flash_attention(q, k, v, causal=True).backward(do)

def loop(n, per_step_events):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        step()
        if per_step_events:
            ev[i + 1].record()
    if not per_step_events:
        ev[n].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[n]) / n * 1e3

for _ in range(400):
    step()
for tag, pse in (("events at the ends only", False), ("an event after every step", True), ("events at the ends only", False), ("an event after every step", True)):
    print(f"{tag:28s}: {loop(100, pse):7.1f} us per step")
big = [torch.randn(4, 32, 4096, D, device=dev, dtype=torch.bfloat16) for _ in range(8)]
x = sum(float(b[0, 0, 0, 0]) for b in big)
del big
for _ in range(400):
    step()
print(f"after holding 8 x 134 MB    : {loop(100, False):7.1f} us per step (ends only), {loop(100, True):7.1f} (every step)")
print("allocator:", {k2: v2 for k2, v2 in torch.cuda.memory_stats().items() if k2 in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current")})
