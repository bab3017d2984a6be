#!/bin/bash
# round 6, GPU session 14: where the window instances' embedded-request tail starts to pay: AULE_HIP_W4_WTAIL in {0 (always), 2, 4 (default), 8, 99 (never)} over a
# row of window lengths, same box; then the window suite once more on the tree.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s14; mkdir -p $O
for w in 4 0 2 8 99 4; do AULE_HIP_W4_WTAIL=$w timeout 300 python tools/window_tail_ab.py 2>&1 | grep -v amdgpu; done > $O/window_tail_ab.txt 2>&1
cat $O/window_tail_ab.txt
( timeout 900 python -m pytest tests/test_gpu_window.py -x -q -m gpu ) > $O/pytest_window.txt 2>&1; tail -3 $O/pytest_window.txt
