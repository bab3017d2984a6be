#!/bin/bash
# tools/cb_dkv4.sh [prev.so] -- the one-wave-per-SIMD dK/dV kernel forced on: timing + checksums (tools/cbench.cpp) of the in-tree library
# (and of a previous build, same box), then the timeline build's cycles per iteration (build/variants/libaule_dbg.so)
L=aule-attention_amd/aule/lib/libaule.so
export AULE_HIP_BWD_DKV=new
for lib in $L $1; do
  echo "== $lib"
  timeout 60 build/cbench $lib bwd 4 32 8 2048 2048 128 bf16 1 40 15 | grep -v " o:\| dq:"
  timeout 60 build/cbench $lib bwd 4 32 8 4096 4096 128 bf16 1 20 8 | grep -v " o:\| dq:"
  timeout 60 build/cbench $lib bwd 2 16 16 4096 4096 128 bf16 0 20 8 | grep -v " o:\| dq:"
done
AULE_TL=dkv4 timeout 60 build/cbench build/variants/libaule_dbg.so tl 4 32 8 2048 2048 128 bf16 1 | tail -2
