#!/bin/bash
# Where does the V-phase's staging time go?  Timing-only builds of the tile stream (results are garbage): without the LDS
# staging writes, without the global K/V loads, without both -- all with the online softmax (no verdict, no second stream).
cd "$(dirname "$0")/.."
export AULE_HIP_FWD_SOFTMAX=classic
for v in base nowrite noload neither; do
  lib=build/variants/libaule_x$v.so
  [ -f $lib ] || { echo "missing $lib"; continue; }
  echo "== $v"
  AULE_LIBRARY_PATH=$PWD/$lib timeout 120 python tools/ps_check.py bench 2>&1 < /dev/null | grep -v amdgpu | tail -8
done
