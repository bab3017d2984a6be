#!/bin/bash
# round 5, GPU session 24: cycle ablation of the D = 64 dQ stream (tools/gen_dq4.py DQ4_X; results are garbage, time is not) + one more fuzz slice with the
# one-wave-per-SIMD pair forced on the tree with the trimmed dK/dV loop
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s24; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new
for rep in 1 2; do
for v in dqx_base dqx_novalu dqx_nolds dqx_nodma dqx_nodq dqx_all; do
  export AULE_LIBRARY_PATH=$R/build/variants/libaule_$v.so
  echo "== $v (rep $rep)"; timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B" | head -4
done
done > $O/d64_dq_ablation.txt 2>&1
cat $O/d64_dq_ablation.txt | cut -c1-100
unset AULE_LIBRARY_PATH
timeout 900 python tools/fuzz_parity.py 300 31 > $O/fuzz_new_31.txt 2>&1; tail -4 $O/fuzz_new_31.txt
