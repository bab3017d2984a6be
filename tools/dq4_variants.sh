#!/bin/bash
# tools/dq4_variants.sh -- timing ablations of the one-wave-per-SIMD dQ kernel's stream (DQ4_X of tools/gen_dq4.py; results are
# garbage, only the time counts): builds build/variants/libaule_dq4x_<name>.so here (no GPU), tools/dq4_variants_run.sh times them
R=/root/repo; C=$R/aule-attention_amd/csrc
mkdir -p $R/build/vobj $R/build/variants
[ -d $R/build/obj_dbg ] || (cd $C && make dbg -j8 > /dev/null)
for n in base nobar novalu nolds nodma nodq nowait novalu,nolds novalu,nolds,nodma,nobar; do
  f=$(echo $n | tr ',' '_')
  DQ4_X=$([ $n = base ] && echo "" || echo $n) DQ4_OUT=$R/build/vobj/dq4_asm_$f.inc python3 $R/tools/gen_dq4.py > /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -DAULE_DEBUG_HOOKS -DDQ4_ASM_INC="\"$R/build/vobj/dq4_asm_$f.inc\"" -I$C -c $C/fa_bwd_dq4_gfx950.hip -o $R/build/vobj/dq4_x_$f.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_dq4x_$f.so $(ls $R/build/obj_dbg/*.o | grep -v "/fa_bwd_dq4_gfx950.o") $R/build/vobj/dq4_x_$f.o -Wl,--no-undefined -Wl,-soname,libaule.so
  echo built $f
done
