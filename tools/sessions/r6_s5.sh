#!/bin/bash
# round 6, GPU session 5: the D = 64 dK/dV instance with TWO key blocks per wave (fa_bwd_dkv4_kernel_d64k2; tools/gen_bw4.py Cfg2): first parity run
# (a quick forced-on slice first, so that a hang or garbage shows before the long legs), then the whole backward suites, then A/B timings k2 off / on.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s5; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new
( AULE_HIP_BWD_DKV_K2=1 timeout 600 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu -k "vs_oracle and 64" ) > $O/pytest_k2_quick.txt 2>&1; tail -15 $O/pytest_k2_quick.txt
for k in 0 1; do echo "== AULE_HIP_BWD_DKV_K2=$k"; AULE_HIP_BWD_DKV_K2=$k timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B"; done > $O/d64_ab.txt 2>&1
for k in 0 1; do echo "== AULE_HIP_BWD_DKV_K2=$k"; AULE_HIP_BWD_DKV_K2=$k timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B" | head -4; done >> $O/d64_ab.txt 2>&1
cat $O/d64_ab.txt
unset AULE_HIP_BWD_MODE AULE_HIP_BWD_DKV AULE_HIP_BWD_DQ
( time timeout 2400 python -m pytest tests/test_gpu_bwd.py -x -q -m gpu ) > $O/pytest_bwd.txt 2>&1; tail -30 $O/pytest_bwd.txt | cut -c1-300
