#!/bin/bash
# round 5, GPU session 7: fp32 backward small-grid split (parity + A/B), B = 1 backward in both modes (does the spill mode pay when dS fits
# the Infinity Cache?), whole GPU suite.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s7; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
for sp in 1 0; do echo "== AULE_HIP_F32_SPLIT=$sp"; AULE_HIP_F32_SPLIT=$sp timeout 200 python tools/f32_bench.py 2>&1 | grep -E "S512|S256|bwd" ; done | tee $O/f32_split_ab.txt
cat > /tmp/bwd_b1.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
def bwd(B, Hq, Hkv, S, D=128, dt=torch.bfloat16):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, True, sc)
    f = lambda: at.bwd_raw(q, k, v, out, do, lse, True, sc)
    for _ in range(100): f()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 40)
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} S{S} D{D}: {best*1e3:.1f} us", flush=True)
for a in ((1, 32, 32, 2048), (1, 32, 8, 2048), (2, 32, 32, 1024), (1, 32, 32, 4096), (1, 16, 16, 2048, 64)): bwd(*a)
PY
for mode in recompute spill; do echo "== AULE_HIP_BWD_MODE=$mode"; AULE_HIP_BWD_MODE=$mode timeout 200 python /tmp/bwd_b1.py 2>&1 | grep bwd; done | tee $O/bwd_b1_modes.txt
