#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=.. ..."  -- A/B builds: recompile ONE kernel file with extra flags and link it
# with the other (default) objects into build/variants/libaule_NAME.so; select it at run time with AULE_LIBRARY_PATH.
set -e
NAME=$1; FILE=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/aule-attention_amd/csrc
OBJ=$R/build/obj
mkdir -p $R/build/variants $R/build/vobj
(cd $C && make -s >/dev/null)
STEM=$(basename $FILE .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS -c $C/$FILE -o $R/build/vobj/${STEM}_$NAME.o
OBJS=$(ls $OBJ/*.o | grep -v "/${STEM}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_$NAME.so $OBJS $R/build/vobj/${STEM}_$NAME.o -Wl,--no-undefined -Wl,-soname,libaule.so
echo built build/variants/libaule_$NAME.so
