#!/bin/bash
# round 6, GPU session 18: the tree (negq read per part, not held) -- parity of the forward suites once more; same-box A/B of the metric shapes' forward against the
# library of session 8 (build/variants/libaule_r6s8.so = commit 244ace2: before the window instances and the negative-scale path): did the headline kernel move?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s18; mkdir -p $O
( timeout 1800 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_splitkv.py tests/test_gpu_rope.py tests/test_gpu_window.py -x -q -m gpu ) > $O/pytest_fwd.txt 2>&1; tail -3 $O/pytest_fwd.txt | cut -c1-300
for rep in 1 2 3; do
  for lib in r6s8 tree; do
    if [ $lib = tree ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
    echo "== $lib (rep $rep)"; timeout 300 python tools/fwd_check.py bench w4 2>&1 | grep -E "S4096 D128 causal=1 lse=1|Hkv8 S2048|S8192 D128|S16384|B16"
  done
done > $O/fwd_ab.txt 2>&1
unset AULE_LIBRARY_PATH
cat $O/fwd_ab.txt | cut -c1-150
