#!/bin/bash
# round 6, GPU session 9: the sliding-window instances of the one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip WIN: part = the block's visible tile
# range, waves that start late, two-sided mask variants; generic bodies only): parity (quick slice first, under a short timeout: a hang must show
# early), the window / rope / bottom-right / capi suites, then timings against the ping-pong route (AULE_HIP_W4_WINDOW=0).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s9; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_window.py -x -q -m gpu -k "2048-2048-128-True-256 or 1536 or large_logits" ) > $O/pytest_quick.txt 2>&1; tail -15 $O/pytest_quick.txt | cut -c1-300
( timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_rope.py -x -q -m gpu ) > $O/pytest_window.txt 2>&1; tail -15 $O/pytest_window.txt | cut -c1-300
for w in 1 0; do echo "== AULE_HIP_W4_WINDOW=$w"; AULE_HIP_W4_WINDOW=$w timeout 300 python tools/window_bench.py 2>&1 | grep -v amdgpu; done > $O/window_bench.txt 2>&1
cat $O/window_bench.txt | cut -c1-220
( timeout 1500 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_capi.py tests/test_gpu_bottom_right.py tests/test_gpu_fwd_variants.py -x -q -m gpu ) > $O/pytest_fwd.txt 2>&1; tail -5 $O/pytest_fwd.txt | cut -c1-300
