#!/bin/bash
# round 5, GPU session 14: causal sliding windows on the one-wave-per-SIMD backward pair (dK/dV: stream clipped at the window, mask with the
# window bound; dQ: stream starts at the first visible block, two-sided mask): parity (window suite by itself, then the three-mode legs),
# then the README's window shapes against the predecessors, same box.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_window.py -m gpu -q --maxfail=6 > $O/pytest_window.txt 2>&1; tail -6 $O/pytest_window.txt
AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new AULE_HIP_BWD_MODE=recompute timeout 900 python -m pytest tests/test_gpu_window.py -m gpu -q --maxfail=6 > $O/pytest_window_forced.txt 2>&1; tail -6 $O/pytest_window_forced.txt
for leg in new old; do echo "== backward kernels: $leg"; AULE_HIP_BWD_DKV=$leg AULE_HIP_BWD_DQ=$leg timeout 300 python tools/window_bench.py 2>&1 | grep window; done | tee $O/window_bench_ab.txt
