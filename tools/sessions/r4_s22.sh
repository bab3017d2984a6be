#!/bin/bash
# round 4, GPU session 22: the one marginal fuzz exceedance again under both backward pairs (dQ is bit-identical between them, so the
# error must be); more random configurations with the new pair forced, another seed, and with the default rules
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s22; mkdir -p $O
for m in old new; do echo "## $m"; AULE_HIP_BWD_DKV=$m AULE_HIP_BWD_DQ=$m timeout 120 python tools/fuzz_parity.py one bf16 1 4 2 31 500 128 br -1 0.3 1001 2>&1 | grep route; done
AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new timeout 900 python tools/fuzz_parity.py 400 11 > $O/fuzz_new_11.txt 2>&1; grep -v amdgpu.ids $O/fuzz_new_11.txt | tail -4
timeout 900 python tools/fuzz_parity.py 300 12 > $O/fuzz_default_12.txt 2>&1; grep -v amdgpu.ids $O/fuzz_default_12.txt | tail -4
