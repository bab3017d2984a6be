#!/bin/bash
# round 6, GPU session 12: fuzz of the window instances (tools/fuzz_parity.py window: 300 draws, two seeds), the general fuzz (its window draws now land
# on route 8 too), then the WHOLE GPU suite on the tree.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s12; mkdir -p $O
( timeout 1200 python tools/fuzz_parity.py window 200 0; timeout 1200 python tools/fuzz_parity.py window 200 1 ) > $O/fuzz_window.txt 2>&1; tail -12 $O/fuzz_window.txt | cut -c1-300
( timeout 900 python tools/fuzz_parity.py 300 6 ) > $O/fuzz_general.txt 2>&1; tail -5 $O/fuzz_general.txt | cut -c1-300
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt | cut -c1-400
