#!/usr/bin/env python3
"""HBM rate of the RoPE pass (csrc/rope_gfx950.hip): algorithmic bytes = 2 * numel * sizeof(T) (read x, write x';
the [S, D/2] fp32 table rows are shared by all heads and stay in L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
from aule import _torch as at

def run(B, H, S, D, dt, layout, inplace):
    x = torch.randn(B, H, S, D, device="cuda", dtype=dt)
    cos, sin = aule.precompute_rope_frequencies(S, D, device="cuda")
    out = x if inplace else torch.empty_like(x)
    for _ in range(5): at.rope_raw(x, cos, sin, layout, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): at.rope_raw(x, cos, sin, layout, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gb = 2 * x.numel() * x.element_size() / 1e9
    print(f"  rope B{B} H{H} S{S} D{D} {str(dt)[6:]} {layout}{' in-place' if inplace else ''}: {us:.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)

if __name__ == "__main__":
    run(4, 32, 4096, 128, torch.bfloat16, "half", False)
    run(4, 32, 4096, 128, torch.bfloat16, "half", True)
    run(4, 32, 4096, 128, torch.bfloat16, "interleaved", False)
    run(4, 8, 4096, 128, torch.bfloat16, "half", False)
    run(4, 32, 4096, 128, torch.float32, "half", False)
    run(4, 32, 2048, 64, torch.float16, "half", False)
