#!/bin/bash
# round 5, GPU session 11: randomised parity with the 5-matmul backward forced onto everything it can run (two seeds) and under the default
# dispatch (un-paired launches, fp32 pieces, auto mode), smoke(), the whole GPU suite on the final tree.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s11; mkdir -p $O
AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 900 python tools/fuzz_parity.py 300 11 > $O/fuzz_spill_11.txt 2>&1; tail -3 $O/fuzz_spill_11.txt
AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 900 python tools/fuzz_parity.py 300 12 > $O/fuzz_spill_12.txt 2>&1; tail -3 $O/fuzz_spill_12.txt
timeout 900 python tools/fuzz_parity.py 300 13 > $O/fuzz_default_13.txt 2>&1; tail -3 $O/fuzz_default_13.txt
timeout 600 python tools/fuzz_parity.py split 60 5 > $O/fuzz_split.txt 2>&1; tail -2 $O/fuzz_split.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 1700 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
