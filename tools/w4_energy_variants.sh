#!/bin/bash
# Energy ablations of the one-wave-per-SIMD forward (round 4): at the socket power limit a launch's time IS its energy / 1400 W, so the time of a
# variant stream with one kind of activity removed (tools/gen_w4.py W4_X=...; results are garbage) prices that activity in joules.
#   tools/w4_energy_variants.sh build            -> build/variants/libaule_w4x_<name>.so
#   tools/w4_energy_variants.sh run <seconds>    -> power_trace legs of every variant on C2 (run on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/aule-attention_amd/csrc
VARIANTS=${W4_VARIANTS:-"base noexp novalu nolds nodma novalu,nolds,nodma"}
if [ "$1" = "build" ]; then
  mkdir -p $R/build/variants $R/build/vobj
  for v in $VARIANTS; do
    n=$(echo $v | tr ',' '_')
    W4_X=$(echo $v | sed -e 's/base//') W4_OUT=$R/build/vobj/w4_asm_$n.inc python3 $R/tools/gen_w4.py >/dev/null
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -Wno-unused-value -DW4_X_NOVERDICT \
       -DW4_ASM_INC="\"$R/build/vobj/w4_asm_$n.inc\"" -I$C -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/fa_fwd_w4_gfx950_x_$n.o &
  done
  wait
  for v in $VARIANTS; do
    n=$(echo $v | tr ',' '_')
    OBJS=$(ls $R/build/obj/*.o | grep -v "/fa_fwd_w4_gfx950.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_w4x_$n.so $OBJS $R/build/vobj/fa_fwd_w4_gfx950_x_$n.o -Wl,--no-undefined -Wl,-soname,libaule.so
    echo "built $n"
  done
else
  cd $R
  S=${2:-1.5}
  for v in $VARIANTS; do
    n=$(echo $v | tr ',' '_')
    timeout 120 build/power_trace build/variants/libaule_w4x_$n.so fwd 4 32 32 4096 4096 128 bf16 1 $S 1.0 w4x_$n
    timeout 120 build/power_trace build/variants/libaule_w4x_$n.so fwd 4 32 32 4096 4096 128 bf16 0 $S 1.0 w4x_nc_$n
  done
fi
