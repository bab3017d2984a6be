#!/bin/bash
# round 6, GPU session 22: a long fuzz of the final tree, every kind, fresh seeds (nothing else to measure: the GPU budget of the round is otherwise spent on records).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s22; mkdir -p $O
{
for s in 21 22; do timeout 1200 python tools/fuzz_parity.py 300 $s; done
timeout 900 python tools/fuzz_parity.py split 120 4
timeout 1500 python tools/fuzz_parity.py long 40 5
for s in 31 32; do timeout 900 python tools/fuzz_parity.py window 200 $s; done
timeout 600 python tools/fuzz_parity.py paged 200 6
timeout 600 python tools/fuzz_parity.py rope 200 7
for m in spill; do AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 900 python tools/fuzz_parity.py 200 23; done
AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new AULE_HIP_BWD_DKV_K2=1 timeout 900 python tools/fuzz_parity.py 200 24
} 2>&1 | grep -v amdgpu > $O/fuzz.txt
grep -c FAIL $O/fuzz.txt; grep -E "configurations|FAIL" $O/fuzz.txt | cut -c1-260
