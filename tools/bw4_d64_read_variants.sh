#!/bin/bash
# tools/bw4_d64_read_variants.sh -- A/B builds of the D = 64 dK/dV stream's LDS read placement (tools/gen_bw4.py: BW4_TR64 = early: the transpose reads of a
# block all in phase-1 statements 0 and 1; BW4_RM64 = a,b,c,d: the row-major fragment reads per phase-2 statement).  Builds build/variants/libaule_rd_<name>.so (no GPU).
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aule-attention_amd/csrc
mkdir -p $R/build/vobj $R/build/variants
(cd $C && make -s -j8 > /dev/null)
build() {   # name, BW4_TR64, BW4_RM64
  BW4_TR64=$2 BW4_RM64=$3 BW4_OUT=$R/build/vobj/bw4_asm_rd_$1.inc python3 $R/tools/gen_bw4.py > /dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -DBW4_ASM_INC="\"$R/build/vobj/bw4_asm_rd_$1.inc\"" -I$C -c $C/fa_bwd_dkv4_gfx950.hip -o $R/build/vobj/dkv4_rd_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_rd_$1.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dkv4_gfx950.o") $R/build/vobj/dkv4_rd_$1.o -Wl,--no-undefined -Wl,-soname,libaule.so
  echo built rd_$1
}
build tr late 2,2,2,2 &
build tre early 2,2,2,2 &
build rm44 late 4,4,0,0 &
build both early 4,4,0,0 &
build both8 early 8,0,0,0 &
wait
