#!/bin/bash
# round 5, GPU session 10: the dQ = dS K kernel un-paired on half-empty grids: parity (backward suites, all modes) + B = 1 timings
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s10; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bottom_right.py -m gpu -q --maxfail=5 > $O/pytest_bwd.txt 2>&1; tail -4 $O/pytest_bwd.txt
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
for a in ((1, 32, 32, 2048), (2, 32, 32, 1024), (1, 16, 16, 2048, 64), (1, 32, 32, 2048, 64), (4, 16, 16, 1024)): bwd(*a)
PY
for mode in recompute auto; do echo "== mode $mode"; if [ $mode = auto ]; then unset AULE_HIP_BWD_MODE; else export AULE_HIP_BWD_MODE=$mode; fi; timeout 200 python /tmp/bwd_b1.py 2>&1 | grep bwd; done | tee $O/bwd_b1_modes.txt
