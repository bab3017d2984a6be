#!/usr/bin/env python3
"""Triage of the sliding-window path: forward (fp32 goldens from the reference kernel, bf16/fp16 vs the fp64
oracle) and backward vs the oracle.  Not a test; prints one line per case."""
import glob, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
import numpy as np, torch, oracle
from aule import _torch as at
DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/win_*.npz"))):
    z = np.load(f); W = int(z["window"]); causal = bool(z["causal"]); D = z["q"].shape[-1]
    q, k, v = (torch.from_numpy(z[n]).cuda() for n in "qkv")
    out, lse = at.fwd_raw(q, k, v, causal, 1 / math.sqrt(D), window=W)
    print(os.path.basename(f), "out err %.2e lse err %.2e" % (np.abs(out.cpu().numpy() - z["out"]).max(), np.abs(lse.cpu().numpy() - z["lse"]).max()), flush=True)
do_bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
for dt, B, Hq, Hkv, Sq, Sk, D, causal, W in [("bf16", 1, 4, 2, 512, 512, 128, True, 100), ("bf16", 1, 2, 2, 1000, 1000, 128, True, 300),
                                             ("bf16", 1, 2, 1, 333, 500, 64, False, 77), ("fp16", 1, 2, 2, 700, 700, 64, True, 64),
                                             ("bf16", 2, 2, 2, 2048, 2048, 128, True, 512), ("fp32", 1, 2, 2, 300, 300, 32, True, 50),
                                             ("bf16", 1, 2, 2, 600, 200, 128, True, 64), ("bf16", 1, 2, 2, 256, 256, 128, True, 1)]:
    rng = np.random.RandomState(5)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32)).to(DT[dt])
    q, k, v, do = mk(B, Hq, Sq, D), mk(B, Hkv, Sk, D), mk(B, Hkv, Sk, D), mk(B, Hq, Sq, D)
    sc = 1 / math.sqrt(D)
    qc, kc, vc, dc = (x.cuda() for x in (q, k, v, do))
    out, lse = at.fwd_raw(qc, kc, vc, causal, sc, window=W)
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal, None, W)
    o = out.float().cpu().numpy(); l = lse.cpu().numpy()
    fin = np.isfinite(rl)
    msg = "%s B%d Hq%d Hkv%d Sq%d Sk%d D%d causal=%d W=%d: out err %.2e lse err %.2e nan=%d" % (
        dt, B, Hq, Hkv, Sq, Sk, D, causal, W, np.abs(o - ref).max(), np.abs(l[fin] - rl[fin]).max(), int(np.isnan(o).sum()))
    if do_bwd:
        dq, dk, dv = at.bwd_raw(qc, kc, vc, out, dc, lse, causal, sc, window=W)
        rq, rk, rv = oracle.bwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), do.float().numpy(), causal, None, W)
        msg += " | dq %.2e dk %.2e dv %.2e (ref max %.1f %.1f %.1f) nan=%d" % (
            np.abs(dq.float().cpu().numpy() - rq).max(), np.abs(dk.float().cpu().numpy() - rk).max(), np.abs(dv.float().cpu().numpy() - rv).max(),
            np.abs(rq).max(), np.abs(rk).max(), np.abs(rv).max(), int(torch.isnan(dq).sum() + torch.isnan(dk).sum() + torch.isnan(dv).sum()))
    print(msg, flush=True)
