#!/bin/bash
# round 6, GPU session 15: window instances -- the block index of an item rotated per head (load balance of the heads' first, short blocks): parity (window suite,
# window fuzz), timings.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s15; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_rope.py -x -q -m gpu ) > $O/pytest_window.txt 2>&1; tail -3 $O/pytest_window.txt
( timeout 900 python tools/fuzz_parity.py window 200 3 ) > $O/fuzz_window.txt 2>&1; tail -2 $O/fuzz_window.txt
for w in 1 1; do timeout 300 python tools/window_bench.py 2>&1 | grep -v amdgpu; done > $O/window_bench.txt 2>&1; cut -c1-150 $O/window_bench.txt
timeout 300 python tools/window_tail_ab.py 2>&1 | grep -v amdgpu > $O/window_tail.txt; cat $O/window_tail.txt
