#!/bin/bash
# tools/cb_split.sh -- dQ kernel and dK/dV kernel timed separately (debug library, AULE_DBG_BWD_ONLY) through tools/cbench.cpp
L=build/variants/libaule_dbg.so
for shape in "4 32 8 2048 2048 128 bf16 1" "4 32 8 4096 4096 128 bf16 1"; do
  for only in dq dkv all; do
    for mode in new old; do
      [ $only = dq ] && [ $mode = old ] && continue
      echo -n "only=$only dkv=$mode: "
      if [ $only = all ]; then unset AULE_DBG_BWD_ONLY; else export AULE_DBG_BWD_ONLY=$only; fi
      AULE_HIP_BWD_DKV=$mode timeout 60 build/cbench $L bwd $shape 10 3 10 | head -1
    done
  done
done
