#!/bin/bash
# tools/dqs_variants.sh -- timing variants of the dQ = dS K kernel (fa_bwd_dqs_gfx950.hip; results of the X variants are garbage):
# builds build/variants/libaule_dqs_<name>.so here (no GPU)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/aule-attention_amd/csrc
mkdir -p $R/build/vobj $R/build/variants
(cd $C && make -s -j8 > /dev/null)
build() {  # name flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm $2 -I$C -c $C/fa_bwd_dqs_gfx950.hip -o $R/build/vobj/dqs_$1.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_dqs_$1.so $(ls $R/build/obj/*.o | grep -v "/fa_bwd_dqs_gfx950.o") $R/build/vobj/dqs_$1.o -Wl,--no-undefined -Wl,-soname,libaule.so
  echo built dqs_$1
}
build nomfma "-DDQS_X_NOMFMA"
build seq "-DDQS_X_SEQ"
build seq_nomfma "-DDQS_X_SEQ -DDQS_X_NOMFMA"
build nods "-DDQS_X_NODS"
build a3 "-DDQS_AHEAD=3 -DDQS_SLOTS=4"
build a2 "-DDQS_AHEAD=2 -DDQS_SLOTS=3"
