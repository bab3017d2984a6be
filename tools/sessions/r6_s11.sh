#!/bin/bash
# round 6, GPU session 11: window instances -- embedded tail only behind >= 4 whole tiles; left-edge-only mask variants (SM 6; windows >= 2 key tiles).
# Quick parity slice under a short timeout, the window / rope / bottom-right suites, timings vs the ping-pong route.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s11; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_window.py -x -q -m gpu -k "2048-2048-128-True-256 or 1536 or large_logits or 129" ) > $O/pytest_quick.txt 2>&1; tail -15 $O/pytest_quick.txt | cut -c1-300
( timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_rope.py tests/test_gpu_bottom_right.py -x -q -m gpu ) > $O/pytest_window.txt 2>&1; tail -15 $O/pytest_window.txt | cut -c1-300
for w in 1 0 1; do echo "== AULE_HIP_W4_WINDOW=$w"; AULE_HIP_W4_WINDOW=$w timeout 300 python tools/window_bench.py 2>&1 | grep -v amdgpu; done > $O/window_bench.txt 2>&1
cat $O/window_bench.txt | cut -c1-150
