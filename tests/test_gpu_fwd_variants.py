"""Parity of the NON-default forward kernels / softmax forms (selected by environment variables that the
library reads once per process, hence one subprocess per variant):

  AULE_HIP_FWD_KERNEL=pp        every tiled problem on the ping-pong kernel (two waves per SIMD, one workgroup per Q-block pair):
                                the predecessor of the one-wave-per-SIMD kernel and still the kernel behind window / short
                                shapes, D = 32 and negative scales
  AULE_HIP_FWD_SOFTMAX=classic  online softmax only (no fixed-reference pass), on the ping-pong kernel (the one-wave-per-SIMD
                                kernel has no online form and is skipped)
  AULE_HIP_FWD_SPLIT=0          small grids without the key-range split (route 8 instead of route 7)
(The two-waves-per-SIMD tile stream of rounds 2-3, AULE_HIP_FWD_KERNEL=ps, was retired in round 4.)

Each variant runs the same seeded cases against the fp64 oracle, including a large-logit case that the
fixed-reference pass must hand over to the online form (its row sums leave the safe range).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, math, os, sys
sys.path.insert(0, os.path.join(%(root)r, "aule-attention_amd")); sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
import oracle
from aule import _torch as at
from util import fwd_tol, LSE_TOL
res = []
cases = [  # B, Hq, Hkv, Sq, Sk, causal, magnitude, spike
    (1, 2, 2, 64, 64, True, 1.0, 0), (1, 2, 2, 300, 300, True, 1.0, 0), (2, 4, 1, 1024, 1024, True, 1.0, 0),
    (1, 2, 2, 200, 333, False, 1.0, 0), (1, 2, 2, 777, 130, False, 1.0, 0), (1, 4, 2, 512, 512, True, 1.0, 0),
    (1, 2, 2, 512, 512, True, 6.0, 0), (1, 2, 2, 512, 512, False, 12.0, 0),
    (2, 40, 8, 1280, 1280, True, 1.0, 0),    # 240 items with an unpaired middle block: every workgroup of the stream walks a list
    # a key whose logit is ~160 (log2) above everything in the first tile, in every other head: the fixed-reference pass
    # must fail its range verdict for those Q blocks in the middle of the part lists and the second, sparse stream
    # (online softmax) must repair exactly them
    (4, 16, 16, 1024, 1024, True, 1.0, 1), (1, 8, 8, 2048, 2048, False, 1.0, 1),
]
for (B, Hq, Hkv, Sq, Sk, causal, mag, spike) in cases:
    rng = np.random.RandomState(7)
    mk = lambda *s: torch.from_numpy((rng.randn(*s) * mag).astype(np.float32)).to(torch.bfloat16)
    q, k, v = mk(B, Hq, Sq, 128), mk(B, Hkv, Sk, 128), mk(B, Hkv, Sk, 128)
    if spike:
        k[:, ::2, Sk - 120, :] = (40.0 * q[:, ::2, Sq - 70, :].float()).to(torch.bfloat16)
    out, lse = at.fwd_raw(q.cuda(), k.cuda(), v.cuda(), causal, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    ref, rl = oracle.fwd_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal)
    o = out.float().cpu().numpy()
    atol, rtol = fwd_tol("bf16", float(v.float().abs().max()))
    bad = int((np.abs(o - ref) > atol + rtol * np.abs(ref)).sum())
    lbad = int((np.abs(lse.cpu().numpy() - rl) > LSE_TOL["bf16"] * max(1.0, mag) + 1e-5 * np.abs(rl)).sum())
    res.append({"case": [B, Hq, Hkv, Sq, Sk, int(causal), mag, spike], "bad": bad, "lse_bad": lbad,
                "nan": int(np.isnan(o).sum()), "max_err": float(np.abs(o - ref).max())})
print("RESULT " + json.dumps(res))
'''


@pytest.mark.parametrize("env", [{}, {"AULE_HIP_FWD_KERNEL": "pp"}, {"AULE_HIP_FWD_SOFTMAX": "classic"},
                                 {"AULE_HIP_FWD_KERNEL": "pp", "AULE_HIP_FWD_SOFTMAX": "classic"}, {"AULE_HIP_FWD_SPLIT": "0"}],
                         ids=["default", "kernel-pp", "softmax-classic", "kernel-pp-softmax-classic", "no-key-split"])
def test_forward_variant_matches_oracle(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    for c in json.loads(line[7:]):
        assert c["nan"] == 0 and c["bad"] == 0 and c["lse_bad"] == 0, c


_BODIES_CHILD = r'''
import hashlib, json, math, os, sys
sys.path.insert(0, os.path.join(%(root)r, "aule-attention_amd"))
import torch
from aule import _torch as at
CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, causal  -- every geometry of the embedded-request flow: short / long parts, ragged, pairs,
           # bottom-right aligned and not, Sq > Sk, GQA, D = 64, small grids (key-range split), no mask at all
    ("bf16", 1, 2, 2, 256, 256, 128, True), ("bf16", 1, 2, 2, 300, 300, 128, True), ("bf16", 2, 4, 1, 1024, 1024, 128, True),
    ("bf16", 1, 3, 3, 1280, 1280, 128, True), ("bf16", 1, 8, 8, 512, 1024, 128, "bottom-right"), ("bf16", 2, 8, 8, 1000, 3000, 128, "bottom-right"),
    ("bf16", 1, 8, 8, 1024, 512, 128, True), ("bf16", 1, 2, 2, 200, 333, 128, False), ("bf16", 1, 2, 2, 777, 260, 128, False),
    ("fp16", 2, 8, 8, 1111, 1111, 64, True), ("fp16", 1, 32, 1, 2048, 2048, 64, False), ("bf16", 4, 32, 8, 2048, 2048, 128, True),
    ("bf16", 1, 8, 8, 4096, 4096, 128, True), ("bf16", 1, 8, 8, 2048, 2048, 128, False),
]
# ... and 18 drawn shapes (fixed seed): ragged lengths, GQA ratios, all three mask modes, both head sizes, 1 to 9 parts per workgroup
import random
rnd = random.Random(4)
for _ in range(18):
    D = rnd.choice((128, 128, 64))
    g = rnd.choice((1, 1, 2, 4))
    Hkv = rnd.choice((1, 2, 3, 5, 8))
    Sq = rnd.choice((257, 300, 512, 640, 777, 1024, 1500, 2048, 2300))
    mode = rnd.choice((True, True, False, "bottom-right"))
    Sk = Sq if mode is True and rnd.random() < 0.7 else max(257, Sq + rnd.choice((-200, 0, 64, 100, 513, 1000)))
    if mode == "bottom-right" and Sk < Sq:
        Sk = Sq + 37
    CASES.append((rnd.choice(("bf16", "fp16")), rnd.choice((1, 2, 3)), Hkv * g, Hkv, Sq, Sk, D, mode))
out = {}
for dtype, B, Hq, Hkv, Sq, Sk, D, causal in CASES:
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=dt, generator=g)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=dt, generator=g)
    o, lse = at.fwd_raw(q, k, v, at.causal_code(causal), 1.0 / math.sqrt(D))
    torch.cuda.synchronize()
    out[str((dtype, B, Hq, Hkv, Sq, Sk, D, str(causal)))] = [hashlib.sha256(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest(),
                                                            hashlib.sha256(lse.cpu().numpy().tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
'''


def test_embedded_request_bodies_are_bit_identical_to_the_generic_ones():
    """Round 4's forward runs step 0, the step in front of a wave's last tile and the last tile through bodies of the plain step's form
    (static ring slots, literal scalar registers, requests in the MFMA gaps).  The arithmetic is the generic bodies': O and LSE of the
    two flows (AULE_HIP_W4_BODIES=generic pins the old one) must agree bit for bit on every geometry (14 chosen ones, 18 drawn ones).  The same switch turns off the seam (the finished
    part's pack inside the next prologue), K_3 / the next part's Q rows / the padding positions' requests in their embedded forms."""
    res = []
    for bodies in ("", "generic"):
        e = {k: v for k, v in os.environ.items() if k != "AULE_HIP_W4_BODIES"}
        if bodies:
            e["AULE_HIP_W4_BODIES"] = bodies
        r = subprocess.run([sys.executable, "-c", _BODIES_CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    assert res[0].keys() == res[1].keys() and len(res[0]) >= 30
    for case in res[0]:
        assert res[0][case] == res[1][case], case
