#!/bin/bash
# round 6, GPU session 7: (a) the two-key-block D = 64 dK/dV stream back on its first arrangement (all arithmetic in phase 1; the split is a generator
# switch now) -- timings; (b) the fp32 small-grid merges by the last arriver WITHOUT device-scope fences (agent-scope stores / loads of the pieces)
# against the merge launches; (c) the whole GPU suite on the tree.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s7; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new
for rep in 1 2; do for k in 0 1; do echo "== AULE_HIP_BWD_DKV_K2=$k"; AULE_HIP_BWD_DKV_K2=$k timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B" | head -4; done; done > $O/d64_ab.txt 2>&1
cat $O/d64_ab.txt
unset AULE_HIP_BWD_MODE AULE_HIP_BWD_DKV AULE_HIP_BWD_DQ
for m in launch kernel launch kernel; do echo "== AULE_HIP_F32_MERGE=$m"; AULE_HIP_F32_MERGE=$m timeout 300 python tools/f32_bench.py 2>&1 | grep -E "S2048 D64 causal=1|S256|S512"; done > $O/f32_bench.txt 2>&1
cat $O/f32_bench.txt
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt | cut -c1-400
