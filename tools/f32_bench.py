#!/usr/bin/env python3
"""fp32 forward / backward timings (the kernels behind the legacy C-ABI and NumPy / fp32 torch input): TFLOP/s against the
157.3 TF f32-MFMA roof.  Select a build with AULE_LIBRARY_PATH."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at


def run(B, Hq, Hkv, S, D, causal, bwd=False):
    q = torch.randn(B, Hq, S, D, device="cuda"); k = torch.randn(B, Hkv, S, D, device="cuda"); v = torch.randn_like(k)
    sc = 1 / math.sqrt(D)
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    do = torch.randn_like(q)
    f = (lambda: at.bwd_raw(q, k, v, out, do, lse, causal, sc)) if bwd else (lambda: at.fwd_raw(q, k, v, causal, sc))
    n = 10
    for _ in range(6):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 4.0 * B * Hq * D * (S * (S + 1) / 2 if causal else S * S) * (2.5 if bwd else 1.0)
    print(f"  fp32 {'bwd' if bwd else 'fwd'} B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} causal={int(causal)}: {ms*1e3:9.1f} us  {fl/ms/1e9:6.1f} TF  "
          f"({fl/ms/1e9/157.3*100:4.1f} % of the f32 MFMA roof)", flush=True)


print("lib:", os.environ.get("AULE_LIBRARY_PATH", "(in-tree)"))
if len(sys.argv) > 1 and sys.argv[1] == "bwd":      # the backward shapes only (A/B of the backward kernels)
    for a in ((4, 32, 32, 2048, 64, True), (4, 32, 32, 2048, 128, True), (4, 32, 32, 2048, 128, False), (2, 32, 8, 4096, 128, True),
              (4, 32, 32, 2048, 64, False), (8, 16, 16, 1024, 64, True)):
        run(*a, True)
    sys.exit(0)
run(1, 8, 8, 2048, 64, True); run(4, 32, 32, 2048, 64, True); run(4, 32, 32, 2048, 128, True); run(4, 32, 32, 2048, 128, False)
run(2, 32, 8, 4096, 128, True); run(1, 8, 8, 256, 64, True)
run(4, 32, 32, 2048, 64, True, True); run(4, 32, 32, 2048, 128, True, True)
# the legacy C-ABI benchmark's shape (the reference's tests/benchmark_attention.zig:18-21: B4 H8 S512 D64, non-causal there) and a GQA one
run(4, 8, 8, 512, 64, False); run(4, 8, 8, 512, 64, False, True); run(4, 8, 8, 512, 64, True, True); run(2, 32, 8, 4096, 128, True, True); run(4, 32, 32, 2048, 128, False, True)
