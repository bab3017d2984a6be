#!/bin/bash
# round 5, GPU session 21: D = 64 dK/dV stream, where the LDS reads of an iteration sit (tools/bw4_d64_read_variants.sh): times + gradient hashes (the variants
# reorder reads only: every hash must equal the committed build's)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s21; mkdir -p $O
export AULE_HIP_BWD_MODE=recompute AULE_HIP_BWD_DKV=new
for rep in 1 2; do
for v in base rd_tr rd_tre rd_rm44 rd_both rd_both8; do
  export AULE_LIBRARY_PATH=$R/build/variants/libaule_$v.so
  echo "== $v (rep $rep)"; timeout 300 python tools/bwd_d64_ab.py 2>&1 | grep "bwd B"
done
done > $O/d64_reads.txt 2>&1
cat $O/d64_reads.txt
