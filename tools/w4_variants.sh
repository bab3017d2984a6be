#!/bin/bash
# Variant libraries of the one-wave-per-SIMD forward for same-box A/Bs and D = 64 timelines:
#   build/variants/libaule_dbg.so          debug hooks (timeline of the bf16 D = 128 kernel)
#   build/variants/libaule_dbg64.so        ... of the bf16 D = 64 kernel
#   build/variants/libaule_pre.so          D = 64 streams in the pre-scaled-Q form (generator: W4_PRE=1), production otherwise
#   build/variants/libaule_dbg64_pre.so    both
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/aule-attention_amd/csrc
HC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -I$C"
make -s -C $C >/dev/null
make -s -C $C dbg >/dev/null
mkdir -p $R/build/variants $R/build/vobj
W4_PRE=1 W4_OUT=$R/build/vobj/w4_asm_pre.inc python3 $R/tools/gen_w4.py >/dev/null
$HC -DAULE_DEBUG_HOOKS -DW4_TL_D64 -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/w4_dbg64.o &
$HC -DW4_NV_D64=52 -DW4_ASM_INC="\"$R/build/vobj/w4_asm_pre.inc\"" -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/w4_pre.o &
$HC -DAULE_DEBUG_HOOKS -DW4_TL_D64 -DW4_NV_D64=52 -DW4_ASM_INC="\"$R/build/vobj/w4_asm_pre.inc\"" -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/w4_dbg64_pre.o &
wait
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/$1 $(ls $R/build/$2/*.o | grep -v "/fa_fwd_w4_gfx950.o") $3 -Wl,--no-undefined -Wl,-soname,libaule.so; }
link libaule_dbg64.so obj_dbg $R/build/vobj/w4_dbg64.o
link libaule_pre.so obj $R/build/vobj/w4_pre.o
link libaule_dbg64_pre.so obj_dbg $R/build/vobj/w4_dbg64_pre.o
ls -la $R/build/variants/libaule_dbg.so $R/build/variants/libaule_dbg64.so $R/build/variants/libaule_pre.so $R/build/variants/libaule_dbg64_pre.so
