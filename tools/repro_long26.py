#!/usr/bin/env python3
"""The one exceedance of session r6_s22's `long` fuzz (draw 26 of seed 5: bf16 B4 H8 S2125 D128 causal window 256, dQ of unit (3, 2) at 7.05e-3 of max|grad| against 6.2e-3):
the same tensors, gradient errors of every (batch, head) unit, under whatever AULE_HIP_W4_WINDOW / AULE_HIP_BWD_* say -- is it the forward route, or the backward's own rounding?"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from aule import _torch as at
B, H, S, D, W, i = 4, 8, 2125, 128, 256, 26
gen = torch.Generator(device="cuda").manual_seed(11000 + i)
tq, tdo = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(2))
tk, tv = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(2))
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(tq, tk, tv, True, sc, window=W)
dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, True, sc, window=W)
torch.cuda.synchronize()
print("AULE_HIP_W4_WINDOW =", os.environ.get("AULE_HIP_W4_WINDOW", "(default)"), " BWD_DKV/DQ =", os.environ.get("AULE_HIP_BWD_DKV", "-"), os.environ.get("AULE_HIP_BWD_DQ", "-"))
worst = {}
for b in range(B):
    for h in range(H):
        f = lambda t: t[b, h:h + 1].float().cpu().numpy()      # [g = 1, S, D]: the query group of the KV head
        f2 = lambda t: t[b, h].float().cpu().numpy()           # [S, D]
        rq, rk, rv = oracle.bwd_head_f64(f(tq), f2(tk), f2(tv), f(tdo), None, True, None, W)
        for name, got, want in (("dq", f(dq), rq), ("dk", f2(dk), rk), ("dv", f2(dv), rv)):
            e = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
            if e > worst.get(name, (0,))[0]: worst[name] = (e, b, h)
print("  worst error / max|grad| over the 32 units:", {k: (round(v[0], 5), v[1], v[2]) for k, v in worst.items()})
# the forward's own error at the same unit
rows = (np.arange(S) + (3 * H + 2) * S).astype(np.int64)
ro, rl = oracle.fwd_rows_f64(tq.float().cpu().numpy(), tk.float().cpu().numpy(), tv.float().cpu().numpy(), rows, True, None, W)
o = out.float().cpu().numpy().reshape(-1, D)[rows]
print("  forward unit (3, 2): out max|err| %.3e, lse max|err| %.3e" % (np.abs(o - ro).max(), np.abs(lse.cpu().numpy().reshape(-1)[rows] - rl).max()))
