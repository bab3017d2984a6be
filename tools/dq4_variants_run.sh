#!/bin/bash
# tools/dq4_variants_run.sh -- on the GPU box: the dQ kernel alone (AULE_DBG_BWD_ONLY=dq) of every ablation library, zero inputs / real data
SH="4 32 8 4096 4096 128 bf16 1"
for L in build/variants/libaule_dq4x_*.so; do
  for amp in 0 1; do
    echo -n "$(basename $L .so | sed 's/libaule_dq4x_//') amp=$amp: "
    CB_AMP=$amp AULE_DBG_BWD_ONLY=dq AULE_HIP_BWD_DQ=new timeout 20 build/cbench $L bwd $SH 6 2 10 | head -1 | sed 's/.*median/median/; s/  min.*//'
  done
done
