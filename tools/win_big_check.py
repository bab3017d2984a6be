#!/usr/bin/env python3
"""A window launch with more items than one round of part tables holds (kW4MaxItems = 64 per workgroup: the grid then has 2 x CUs workgroups), checked against
the fp64 judge on sampled rows and -- run once more with AULE_HIP_W4_WINDOW=0 -- against the ping-pong route through a saved sample."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from aule import _torch as at
from util import fwd_tol, LSE_TOL
B, H, Hkv, S, D, W = 8, 32, 8, 16384, 128, 256           # 64 blocks x 256 heads = 16384 items on 256 CUs: 64 per workgroup ... x 2 with B = 16
if len(sys.argv) > 1: B = int(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(3)
q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=g)
k = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16, generator=g); v = torch.randn(B, Hkv, S, D, device="cuda", dtype=torch.bfloat16, generator=g)
out, lse = at.fwd_raw(q, k, v, True, 1 / math.sqrt(D), window=W)
torch.cuda.synchronize()
assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()
rng = np.random.RandomState(1)
rows = rng.randint(0, B * H * S, size=64).astype(np.int64)
ro, rl = oracle.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), rows, True, None, W)
o = out.float().cpu().numpy().reshape(-1, D)[rows]; gl = lse.cpu().numpy().reshape(-1)[rows]
atol, rtol = fwd_tol("bf16", float(v.float().abs().max()))
bad = int((np.abs(o - ro) > atol + rtol * np.abs(ro)).sum()); lbad = float(np.abs(gl - rl).max())
print(f"B{B} Hq{H} Hkv{Hkv} S{S} D{D} W{W}  window route {os.environ.get('AULE_HIP_W4_WINDOW', '1')}: sampled rows bad {bad}, max|err| {np.abs(o-ro).max():.3e}, lse max|err| {lbad:.2e}")
assert bad == 0 and lbad < LSE_TOL["bf16"]
