#!/bin/bash
# Timing experiments on the one-wave-per-SIMD forward: what does each kind of filler cost next to the MFMAs?  Builds variant
# libraries from differently generated instruction streams (tools/gen_w4.py, W4_X=...; results are garbage) into
# build/variants/libaule_w4x_<name>.so.   tools/w4_experiments.sh build | run
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/aule-attention_amd/csrc
VARIANTS=${W4_VARIANTS:-"base noexp dropexp dropfma dropadd nocvt novalu nolds nodma nolds,nodma novalu,nolds,nodma"}
if [ "$1" = "build" ]; then
  (cd $C && make -s >/dev/null && make -s dbg >/dev/null)
  mkdir -p $R/build/variants $R/build/vobj
  for v in $VARIANTS; do
    n=$(echo $v | tr ',' '_')
    W4_SCALE=$(case $v in *fmac*) echo fmac;; *mulsub*) echo mulsub;; *) echo fma;; esac) W4_X=$(echo $v | sed -e 's/base//' -e 's/fmac//' -e 's/mulsub//' -e 's/^,//' -e 's/,$//') W4_OUT=$R/build/vobj/w4_asm_$n.inc python3 $R/tools/gen_w4.py >/dev/null
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -Wno-unused-value -DW4_X_NOVERDICT \
       -DW4_ASM_INC="\"$R/build/vobj/w4_asm_$n.inc\"" -I$C -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/fa_fwd_w4_gfx950_x_$n.o
    OBJS=$(ls $R/build/obj/*.o | grep -v "/fa_fwd_w4_gfx950.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_w4x_$n.so $OBJS $R/build/vobj/fa_fwd_w4_gfx950_x_$n.o -Wl,--no-undefined -Wl,-soname,libaule.so
    # the same with the timeline hooks (cycle counts: wall time is confounded by the clock the data-dependent power draw allows)
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -Wno-unused-value -DW4_X_NOVERDICT -DAULE_DEBUG_HOOKS \
       -DW4_ASM_INC="\"$R/build/vobj/w4_asm_$n.inc\"" -I$C -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/fa_fwd_w4_gfx950_xd_$n.o
    OBJS=$(ls $R/build/obj_dbg/*.o | grep -v "/fa_fwd_w4_gfx950.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_w4xd_$n.so $OBJS $R/build/vobj/fa_fwd_w4_gfx950_xd_$n.o -Wl,--no-undefined -Wl,-soname,libaule.so
    echo "built $n"
  done
else
  cd $R
  for v in $VARIANTS; do
    n=$(echo $v | tr ',' '_')
    echo "== $n"
    AULE_LIBRARY_PATH=$R/build/variants/libaule_w4x_$n.so timeout 300 python tools/fwd_check.py one bf16 4 32 32 4096 128 0 40 2>&1 | grep "bf16 B4"
    AULE_LIBRARY_PATH=$R/build/variants/libaule_w4x_$n.so timeout 300 python tools/fwd_check.py one bf16 4 32 32 4096 128 1 40 2>&1 | grep "bf16 B4"
    AULE_LIBRARY_PATH=$R/build/variants/libaule_w4xd_$n.so timeout 300 python tools/timeline_w4.py 0 4 32 4096 3 2>&1 | grep -m2 "plain"
  done
fi
