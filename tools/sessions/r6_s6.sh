#!/bin/bash
# round 6, GPU session 6: (a) the two-key-block D = 64 dK/dV stream with key block B's arithmetic moved into phase 2 (parity slice, A/B timings, the
# backward suites); (b) the fp32 forward's small-grid merge by the last arriver inside the kernel (parity: forward / C-ABI / graph suites; timing vs the
# merge launch, AULE_HIP_F32_MERGE=launch).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s6; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new
( AULE_HIP_BWD_DKV_K2=1 timeout 600 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu -k "vs_oracle and 64" ) > $O/pytest_k2_quick.txt 2>&1; tail -5 $O/pytest_k2_quick.txt
for rep in 1 2; do for k in 0 1; do echo "== AULE_HIP_BWD_DKV_K2=$k"; AULE_HIP_BWD_DKV_K2=$k timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B" | head -4; done; done > $O/d64_ab.txt 2>&1
cat $O/d64_ab.txt
unset AULE_HIP_BWD_MODE AULE_HIP_BWD_DKV AULE_HIP_BWD_DQ
( timeout 900 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_capi.py tests/test_gpu_graph.py -x -q -m gpu ) > $O/pytest_fwd.txt 2>&1; tail -5 $O/pytest_fwd.txt
for m in launch kernel launch kernel; do echo "== AULE_HIP_F32_MERGE=$m"; AULE_HIP_F32_MERGE=$m timeout 300 python tools/f32_bench.py 2>&1 | grep -v amdgpu; done > $O/f32_bench.txt 2>&1
cat $O/f32_bench.txt
( time timeout 2400 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu ) > $O/pytest_bwd.txt 2>&1; tail -6 $O/pytest_bwd.txt | cut -c1-300
