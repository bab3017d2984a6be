#!/bin/bash
# round 6, GPU session 17: negative scales on the one-wave-per-SIMD forward (Q fragments negated in registers; c = |scale| log2 e): the forward sweep, key-range
# split, rope, window, C-ABI and variant suites; general fuzz (its scale draws include negative ones), window fuzz; C2 timing unchanged?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s17; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_fwd.py -x -q -m gpu -k "vs_oracle" ) > $O/pytest_quick.txt 2>&1; tail -5 $O/pytest_quick.txt | cut -c1-300
( timeout 2400 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_splitkv.py tests/test_gpu_rope.py tests/test_gpu_window.py tests/test_gpu_capi.py tests/test_gpu_fwd_variants.py tests/test_gpu_graph.py tests/test_gpu_bottom_right.py -x -q -m gpu ) > $O/pytest_fwd.txt 2>&1; tail -5 $O/pytest_fwd.txt | cut -c1-300
( timeout 900 python tools/fuzz_parity.py 300 11; timeout 600 python tools/fuzz_parity.py split 60 1; timeout 600 python tools/fuzz_parity.py window 100 8 ) 2>&1 | grep -v amdgpu > $O/fuzz.txt; tail -6 $O/fuzz.txt | cut -c1-300
python tools/fwd_check.py bench w4 2>&1 | grep -v amdgpu | head -4
