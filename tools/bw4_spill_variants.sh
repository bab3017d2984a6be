#!/bin/bash
# tools/bw4_spill_variants.sh -- A/B builds of the SPILL dK/dV kernel's dS stores (tools/gen_bw4.py: BW4_ST = start | end, BW4_ST_NT = 0 | 1)
# and of the dQ = dS K kernel's LDS-DMA source pattern (DQS_DMA_QUAD).  Builds build/variants/libaule_sp_<name>.so here (no GPU).
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aule-attention_amd/csrc
mkdir -p $R/build/vobj $R/build/variants
(cd $C && make -s -j8 > /dev/null)
for v in start_0 start_1 end_0 end_1; do
  st=${v%_*}; nt=${v#*_}
  BW4_ST=$st BW4_ST_NT=$nt BW4_OUT=$R/build/vobj/bw4_asm_$v.inc python3 $R/tools/gen_bw4.py > /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -DBW4_ASM_INC="\"$R/build/vobj/bw4_asm_$v.inc\"" -I$C -c $C/fa_bwd_dkv4_gfx950.hip -o $R/build/vobj/dkv4_sp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_sp_$v.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dkv4_gfx950.o") $R/build/vobj/dkv4_sp_$v.o -Wl,--no-undefined -Wl,-soname,libaule.so
  echo built sp_$v
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -DDQS_DMA_QUAD=0 -I$C -c $C/fa_bwd_dqs_gfx950.hip -o $R/build/vobj/dqs_noquad.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_sp_noquad.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dqs_gfx950.o") $R/build/vobj/dqs_noquad.o -Wl,--no-undefined -Wl,-soname,libaule.so
echo built sp_noquad
