#!/bin/bash
# round 5, GPU session 22: cycle ablation of the D = 64 dK/dV stream (tools/gen_bw4.py BW4_X: one kind of filler removed; results are garbage, time is not)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s22; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new
for rep in 1 2; do
for v in base x_novalu x_nolds x_nodma x_noscal x_all; do
  export AULE_LIBRARY_PATH=$R/build/variants/libaule_$v.so
  echo "== $v (rep $rep)"; timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B" | head -4
done
done > $O/d64_ablation.txt 2>&1
cat $O/d64_ablation.txt
