#!/bin/bash
# round 5, GPU session 20: dK/dV kernel with fewer compiler scalars per iteration (whole descriptors kept as tuples, the LDS base folded into the
# lane constants, the cursor without its block counter, the masked statements outside the loop body) -- parity + A/B against the committed build
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu > $O/pytest_bwd.txt 2>&1; tail -3 $O/pytest_bwd.txt
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export AULE_LIBRARY_PATH=$R/build/variants/libaule_base.so; else unset AULE_LIBRARY_PATH; fi
    echo "== $lib (rep $rep)"; timeout 300 python tools/bwd_ab.py 2>&1 | grep "bwd B"
  done
done > $O/bwd_ab.txt 2>&1
cat $O/bwd_ab.txt
