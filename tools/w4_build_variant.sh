# usage: build_var.sh name ENV...   -> build/variants/libaule_w4xd_<name>.so (debug hooks) and libaule_w4x_<name>.so
R=/root/repo; C=$R/aule-attention_amd/csrc; n=$1; shift
env "$@" W4_OUT=$R/build/vobj/w4_asm_$n.inc python3 $R/tools/gen_w4.py > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-inline-asm -DAULE_DEBUG_HOOKS -DW4_ASM_INC="\"$R/build/vobj/w4_asm_$n.inc\"" -I$C -c $C/fa_fwd_w4_gfx950.hip -o $R/build/vobj/w4_xd_$n.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libaule_w4xd_$n.so $(ls $R/build/obj_dbg/*.o | grep -v "/fa_fwd_w4_gfx950.o") $R/build/vobj/w4_xd_$n.o -Wl,--no-undefined -Wl,-soname,libaule.so
echo built $n
