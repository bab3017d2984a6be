#!/bin/bash
# tools/cb_bwd.sh <lib.so> -- backward A/B of one library through tools/cbench.cpp (no Python): C3 and two large grids, both dK/dV kernels
L=${1:-aule-attention_amd/aule/lib/libaule.so}
for mode in new old; do
  echo "== AULE_HIP_BWD_DKV=$mode  $L"
  AULE_HIP_BWD_DKV=$mode timeout 60 build/cbench $L bwd 4 32 8 2048 2048 128 bf16 1 40 15
  AULE_HIP_BWD_DKV=$mode timeout 60 build/cbench $L bwd 4 32 8 4096 4096 128 bf16 1 20 8
  AULE_HIP_BWD_DKV=$mode timeout 60 build/cbench $L bwd 2 16 16 4096 4096 128 bf16 0 20 8
done
