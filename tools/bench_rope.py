#!/usr/bin/env python3
"""HBM rate of the RoPE pass (csrc/rope_gfx950.hip): algorithmic bytes = 2 * numel * sizeof(T) (read x, write x';
the [S, D/2] fp32 table rows are shared by all heads and stay in L2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
import aule
from aule import _torch as at

def run(B, H, S, D, dt, layout, inplace, nbuf=1):
    """nbuf > 1 cycles over that many distinct tensors so that the working set (in + out) exceeds the 256 MB
    Infinity Cache: only then is the printed rate an HBM rate.  nbuf = 1 re-uses one buffer and is cache-assisted."""
    xs = [torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(nbuf)]
    cos, sin = aule.precompute_rope_frequencies(S, D, device="cuda")
    outs = xs if inplace else [torch.empty_like(x) for x in xs]
    iters = max(48 // nbuf, 4) * nbuf
    for i in range(nbuf): at.rope_raw(xs[i], cos, sin, layout, out=outs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): at.rope_raw(xs[i % nbuf], cos, sin, layout, out=outs[i % nbuf])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    gb = 2 * xs[0].numel() * xs[0].element_size() / 1e9
    ws = nbuf * (1 if inplace else 2) * xs[0].numel() * xs[0].element_size() / 1e6
    kind = "HBM" if ws > 600 else "cache-assisted"
    print(f"  rope B{B} H{H} S{S} D{D} {str(dt)[6:]} {layout}{' in-place' if inplace else ''} "
          f"[working set {ws:.0f} MB, {kind}]: {us:.1f} us  {gb/us*1e3:.2f} TB/s", flush=True)

if __name__ == "__main__":
    run(4, 32, 4096, 128, torch.bfloat16, "half", False)            # one buffer: sits in the Infinity Cache
    run(4, 32, 4096, 128, torch.bfloat16, "half", False, nbuf=6)    # 1.6 GB working set: the HBM rate
    run(4, 32, 4096, 128, torch.bfloat16, "half", True, nbuf=8)
    run(4, 32, 4096, 128, torch.bfloat16, "interleaved", False, nbuf=6)
    run(4, 32, 4096, 128, torch.float32, "half", False, nbuf=4)
    run(4, 32, 4096, 128, torch.bfloat16, "interleaved", False)
    run(4, 8, 4096, 128, torch.bfloat16, "half", False)
    run(4, 32, 4096, 128, torch.float32, "half", False)
    run(4, 32, 2048, 64, torch.float16, "half", False)
