#!/bin/bash
# tools/bw4_energy_variants.sh [build|run] -- energy / cycle ablation of the recompute pair's instruction streams (the
# forward's twin: tools/w4_energy_variants.sh).  build (no GPU): libaule with ONE kind of filler removed from the dK/dV stream (BW4_X of
# tools/gen_bw4.py) or from the dQ stream (DQ4_X of tools/gen_dq4.py) -> build/variants/libaule_bx_<name>.so; results of those libraries are
# garbage, their time is not.  run (GPU): the whole C3 backward of every library through tools/cbench.cpp on N(0,1) data (at the socket's
# power limit a launch's time is its energy / 1400 W: the difference to `base` prices the activity in joules) and on zeros (2.4 GHz: cycles).
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aule-attention_amd/csrc
if [ "${1:-build}" = build ]; then
  mkdir -p $R/build/vobj $R/build/variants
  (cd $C && make -s -j8 > /dev/null)
  HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -I$C"
  for x in base novalu nolds nodma noscal novalu,nolds,nodma,noscal; do
    f=$(echo $x | tr ',' '_')
    BW4_X=$([ $x = base ] && echo "" || echo $x) BW4_OUT=$R/build/vobj/bx_dkv_$f.inc python3 $R/tools/gen_bw4.py > /dev/null
    $HC -DBW4_ASM_INC="\"$R/build/vobj/bx_dkv_$f.inc\"" -c $C/fa_bwd_dkv4_gfx950.hip -o $R/build/vobj/bx_dkv_$f.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_bx_dkv_$f.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dkv4_gfx950.o") $R/build/vobj/bx_dkv_$f.o -Wl,--no-undefined -Wl,-soname,libaule.so
    echo built bx_dkv_$f
  done
  for x in novalu nolds nodma nodq novalu,nolds,nodma; do
    f=$(echo $x | tr ',' '_')
    DQ4_X=$x DQ4_OUT=$R/build/vobj/bx_dq_$f.inc python3 $R/tools/gen_dq4.py > /dev/null
    $HC -DDQ4_ASM_INC="\"$R/build/vobj/bx_dq_$f.inc\"" -c $C/fa_bwd_dq4_gfx950.hip -o $R/build/vobj/bx_dq_$f.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_bx_dq_$f.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dq4_gfx950.o") $R/build/vobj/bx_dq_$f.o -Wl,--no-undefined -Wl,-soname,libaule.so
    echo built bx_dq_$f
  done
else
  cd $R
  SH="4 32 8 2048 2048 128 bf16 1"
  for L in build/variants/libaule_bx_dkv_base.so $(ls build/variants/libaule_bx_*.so | grep -v dkv_base); do
    for amp in 1 0; do
      echo -n "$(basename $L .so | sed 's/libaule_bx_//') amp=$amp: "
      CB_AMP=$amp AULE_HIP_BWD_MODE=recompute timeout 40 build/cbench $L bwd $SH 12 4 150 | head -1 | sed 's/.*median/median/; s/  min.*//'
    done
  done
fi
