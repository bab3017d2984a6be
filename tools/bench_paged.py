#!/usr/bin/env python3
"""Paged-KV decode throughput at the shape family of the reference's only published absolute numbers
(python/README.md:25-32: "PagedAttention Decode (batch=8)", context 1K/2K/4K/8K -> 34 397 / 20 083 / 10 915 /
5 744 tok/s on MI300X; heads and head_dim are not stated there -- LLaMA-style 32 q / 8 kv heads, D = 128, fp16,
block_size 16 are assumed here).  tok/s = batch / time per decode step (one attention layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule

B, Hq, Hkv, D, bs = 8, 32, 8, 128, 16
print(f"paged decode: batch {B}, heads {Hq}q/{Hkv}kv, head_dim {D}, fp16, block_size {bs}")
for ctx in (1024, 2048, 4096, 8192, 32768):
    nb = ctx // bs
    kc = torch.randn(B * nb, bs, Hkv, D, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    q = torch.randn(B, Hq, D, device="cuda", dtype=torch.float16)
    bt = torch.randperm(B * nb, device="cuda").to(torch.int32).view(B, nb)
    cl = torch.full((B,), ctx, device="cuda", dtype=torch.int32)
    for _ in range(10):
        aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        aule.flash_attention_paged_amd(q, kc, vc, bt, cl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byt = 2 * 2 * B * ctx * Hkv * D
    print(f"  ctx {ctx:6d}: {us:7.1f} us/step  {B / us * 1e6:10.0f} tok/s  K+V read {byt / us / 1e3:7.0f} GB/s")
