#!/bin/bash
# Where does the V-phase's staging time go?  Timing-only builds of the REGISTER-STAGED tile stream (results are garbage):
# without the LDS staging writes, without the global K/V loads, without both -- all with the online softmax (no verdict, no
# second stream).  The experiment that led to the LDS-DMA staging; the variants are built with
#   tools/build_variant.sh xbase    fa_fwd_ps_gfx950.hip "-DAULE_PS_DMA=0"
#   tools/build_variant.sh xnowrite fa_fwd_ps_gfx950.hip "-DAULE_PS_DMA=0 -DAULE_PS_X_NOWRITE"
#   tools/build_variant.sh xnoload  fa_fwd_ps_gfx950.hip "-DAULE_PS_DMA=0 -DAULE_PS_X_NOLOAD"
#   tools/build_variant.sh xneither fa_fwd_ps_gfx950.hip "-DAULE_PS_DMA=0 -DAULE_PS_X_NOWRITE -DAULE_PS_X_NOLOAD"
# Read the result with care: with constant data in LDS the matrix pipes draw less power and the chip clocks higher.
cd "$(dirname "$0")/.."
export AULE_HIP_FWD_SOFTMAX=classic
for v in base nowrite noload neither; do
  lib=build/variants/libaule_x$v.so
  [ -f $lib ] || { echo "missing $lib"; continue; }
  echo "== $v"
  AULE_LIBRARY_PATH=$PWD/$lib timeout 120 python tools/ps_check.py bench 2>&1 < /dev/null | grep -v amdgpu | tail -8
done
