#!/bin/bash
# round 5, GPU session 6: the fp32 forward's key-range pieces for small grids and the un-paired causal launch of the one-wave-per-SIMD
# forward on half-empty grids: parity (whole GPU suite), then A/B on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s6; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
for sp in 1 0; do echo "== AULE_HIP_F32_SPLIT=$sp"; AULE_HIP_F32_SPLIT=$sp timeout 200 python tools/f32_bench.py 2>&1 | grep -E "S512|S256|S2048 D64 causal=1" ; done | tee $O/f32_split_ab.txt
cat > /tmp/fwd_small.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
import aule
def t(B, H, S, D=128, dt=torch.float16, n=200):
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=dt) for _ in range(3))
    f = lambda: aule.flash_attention(q, k, v, causal=True)
    with torch.no_grad():
        for _ in range(300): f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
    fl = 4.0 * B * H * D * S * (S + 1) / 2
    print(f"  fwd B{B} H{H} S{S} D{D} {str(dt)[6:]} causal: {best*1e3:7.1f} us  {fl/best/1e9:7.1f} TF", flush=True)
for a in ((1, 32, 2048), (2, 32, 1024), (1, 16, 2048), (4, 16, 1024), (1, 8, 4096), (2, 64, 512), (1, 32, 2048, 64), (1, 32, 4096)):
    t(*a)
PY
for up in 1 0; do echo "== AULE_HIP_W4_UNPAIR=$up"; AULE_HIP_W4_UNPAIR=$up timeout 200 python /tmp/fwd_small.py 2>&1 | grep fwd; done | tee $O/w4_unpair_ab.txt
