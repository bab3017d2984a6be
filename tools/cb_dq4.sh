#!/bin/bash
# tools/cb_dq4.sh [shape...] -- the one-wave-per-SIMD dQ kernel (AULE_HIP_BWD_DQ=new) against its predecessor: whole backward through
# tools/cbench.cpp, dq checksums must agree bit for bit
L=aule-attention_amd/aule/lib/libaule.so
SH=${*:-"1 4 4 512 512 128 bf16 1"}
for mode in old new; do
  echo "== AULE_HIP_BWD_DQ=$mode"
  AULE_HIP_BWD_DQ=$mode timeout 20 build/cbench $L bwd $SH 5 2 | grep -v " o:"
done
