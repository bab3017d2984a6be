#!/bin/bash
# round 5, GPU session 12: where does the 5-matmul mode stop paying as the dS grows past the Infinity Cache (threshold of the auto rule), and
# does it pay for small GQA problems whose dK/dV grid the rule keeps on the predecessor kernel?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s12; mkdir -p $O
cat > /tmp/bwd_sz.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
def bwd(B, Hq, Hkv, S, D=128, dt=torch.bfloat16):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, True, sc)
    f = lambda: at.bwd_raw(q, k, v, out, do, lse, True, sc)
    for _ in range(80): f()
    torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 30)
    mb = B * Hq * (S // 32) * (S // 32) * 2048 / 2 / 1e6
    print(f"  bwd B{B} Hq{Hq} Hkv{Hkv} S{S} D{D} (touched dS {mb:.0f} MB): {best*1e3:.1f} us", flush=True)
for a in ((1, 32, 32, 2048), (1, 48, 48, 2048), (2, 32, 32, 2048), (3, 32, 32, 2048), (4, 32, 32, 2048), (1, 32, 32, 3072), (2, 32, 32, 2048, 64), (4, 32, 32, 2048, 64),
          (1, 32, 8, 2048), (2, 32, 8, 2048), (1, 32, 4, 2048), (2, 16, 2, 2048), (1, 32, 8, 4096)): bwd(*a)
PY
for mode in recompute spill; do echo "== AULE_HIP_BWD_MODE=$mode"; AULE_HIP_BWD_MODE=$mode timeout 300 python /tmp/bwd_sz.py 2>&1 | grep bwd; done | tee $O/bwd_sizes.txt
echo "== AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new (the grid rule lifted)" | tee -a $O/bwd_sizes.txt
AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 300 python /tmp/bwd_sz.py 2>&1 | grep bwd | tee -a $O/bwd_sizes.txt
