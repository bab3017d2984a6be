#!/bin/bash
# round 6, GPU session 13: the window fuzz again after the fp16 verdict fix (sum of weights >= 1 in the fp16 window instances), three seeds, and the window suite.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s13; mkdir -p $O
( for s in 0 1 2; do timeout 1200 python tools/fuzz_parity.py window 200 $s; done ) > $O/fuzz_window.txt 2>&1; grep -v amdgpu $O/fuzz_window.txt | tail -12 | cut -c1-300
( timeout 900 python -m pytest tests/test_gpu_window.py -x -q -m gpu ) > $O/pytest_window.txt 2>&1; tail -3 $O/pytest_window.txt
AULE_HIP_W4_WINDOW=1 timeout 300 python tools/window_bench.py 2>&1 | grep -v amdgpu > $O/window_bench.txt; cut -c1-150 $O/window_bench.txt
