#!/usr/bin/env python3
"""Diagnostics of the fused query rotation: where and by how much it differs from the two-pass form."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at

torch.manual_seed(0)
for (B, H, S, D, causal) in ((1, 2, 512, 128, 0), (2, 8, 1024, 128, 1), (1, 2, 512, 64, 0)):
    q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device="cuda", dtype=torch.float32) / D))
    ang = torch.arange(S, device="cuda", dtype=torch.float32)[:, None] * inv[None, :]
    sc = 1 / math.sqrt(D)
    for name, cos, sin in (("identity", torch.ones_like(ang), torch.zeros_like(ang)),
                           ("swap(cos=0,sin=1)", torch.zeros_like(ang), torch.ones_like(ang)),
                           ("real", ang.cos().contiguous(), ang.sin().contiguous())):
        qr = at.rope_raw(q, cos, sin)
        a = at.fwd_raw(qr, k, v, causal, sc, want_lse=False)[0]
        a2 = at.fwd_raw(qr, k, v, causal, sc, want_lse=False)[0]
        b = at.fwd_raw(q, k, v, causal, sc, want_lse=False, q_rope=(cos, sin, 0))[0]
        b2 = at.fwd_raw(q, k, v, causal, sc, want_lse=False, q_rope=(cos, sin, 0))[0]
        d = (a.float() - b.float()).abs()
        rows = (d.amax(dim=-1) > 0)
        print(f"B{B} H{H} S{S} D{D} c{causal} {name:18s} equal={torch.equal(a, b)} rerun-equal={torch.equal(a, a2)},{torch.equal(b, b2)} "
              f"max|d|={d.max().item():.3e} differing rows={int(rows.sum())}/{rows.numel()} "
              f"first rows={rows.nonzero()[:6].tolist()}", flush=True)
        if name == "real" and rows.any():
            idx = rows.nonzero()[0].tolist()
            bi, hi, si = idx
            print("   row", idx, "a:", a[bi, hi, si, :6].tolist(), "b:", b[bi, hi, si, :6].tolist())
            per_s = rows.float().mean(dim=(0, 1))
            print("   fraction of differing rows by 32-row group:", [round(x, 2) for x in per_s.view(-1, 32).mean(dim=1).tolist()][:16])
