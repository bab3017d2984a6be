#!/bin/bash
# round 5, GPU session 17: final tree after the windowed backward: whole GPU suite; fuzz with the one-wave-per-SIMD pair forced (windows of every
# size incl. 1 and 7 keys, bottom-right, ragged), with the 5-matmul mode forced, and under the default dispatch
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s17; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new timeout 900 python tools/fuzz_parity.py 400 21 > $O/fuzz_new_21.txt 2>&1; tail -3 $O/fuzz_new_21.txt
AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 900 python tools/fuzz_parity.py 300 22 > $O/fuzz_spill_22.txt 2>&1; tail -2 $O/fuzz_spill_22.txt
timeout 900 python tools/fuzz_parity.py 300 23 > $O/fuzz_default_23.txt 2>&1; tail -2 $O/fuzz_default_23.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
