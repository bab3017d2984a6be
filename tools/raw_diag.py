#!/usr/bin/env python3
"""Why does the fixed-reference softmax lose elements at large logit magnitudes?  Dumps the worst rows of a mag-5 case."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import torch
from aule import _torch as at
torch.manual_seed(0)
B, H, S, D, mag = 1, 4, 1024, 128, 5.0
g = torch.Generator(device="cuda").manual_seed(3)
q, k, v = ((torch.randn(B, H, S, D, device="cuda", generator=g) * mag).to(torch.bfloat16) for _ in range(3))
sc = 1 / math.sqrt(D)
out, lse = at.fwd_raw(q, k, v, True, sc)
s = (q.double() @ k.double().transpose(-1, -2)) * sc
i = torch.arange(S, device="cuda")
s = s.masked_fill(i[None, :] > i[:, None], float("-inf"))
p = torch.softmax(s, dim=-1)
ref = p @ v.double()
err = (out.double() - ref).abs()
vmax = v.float().abs().max().item()
bound = 1e-3 + 2 ** -9 * vmax + 2 ** -8 * ref.abs()
bad = (err > bound)
print("softmax mode", os.environ.get("AULE_HIP_FWD_SOFTMAX", "raw"), "bad elements", int(bad.sum()), "max err", err.max().item(), "vmax", vmax)
# emulate: P rounded to bf16 against the TRUE max (what any exact-softmax kernel with bf16 P does)
pm = torch.exp2((s - s.max(dim=-1, keepdim=True).values) * math.log2(math.e))
emu = (pm.float().to(torch.bfloat16).double() @ v.double()) / pm.sum(-1, keepdim=True)
emu_bf = emu.float().to(torch.bfloat16).double()
print("emulation (bf16 P vs true max, fp64 elsewhere, bf16 O): bad", int(((emu_bf - ref).abs() > bound).sum()), "max err", (emu_bf - ref).abs().max().item())
idx = torch.nonzero(bad)
for b_, h_, r_, c_ in idx[:8].tolist():
    row = s[b_, h_, r_] * math.log2(math.e)
    m0 = row[:64].max().item(); mt = row.max().item(); am = int(row.argmax())
    top = torch.topk(p[b_, h_, r_], 3)
    print(f"  row {r_} col {c_}: out {out[b_,h_,r_,c_].item():.4f} ref {ref[b_,h_,r_,c_].item():.4f} err {err[b_,h_,r_,c_].item():.4f} bound {bound[b_,h_,r_,c_].item():.4f} | "
          f"log2 max tile0 {m0:.1f} row max {mt:.1f} at key {am} | top p {[round(x,4) for x in top.values.tolist()]} keys {top.indices.tolist()} | emu err {abs(emu_bf[b_,h_,r_,c_].item()-ref[b_,h_,r_,c_].item()):.4f}")
