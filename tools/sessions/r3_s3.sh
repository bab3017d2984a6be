R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; V=$R/build/variants; cd $R
W4_TL_D=64 W4_TL_HKV=1 AULE_LIBRARY_PATH=$V/libaule_dbg64.so timeout 200 python tools/timeline_w4.py 0 1 32 16384 0 3 > $O/s2_tl_d64_pre.txt 2>&1; grep -v "^/opt" $O/s2_tl_d64_pre.txt | head -24
W4_TL_D=64 W4_TL_HKV=1 AULE_LIBRARY_PATH=$V/libaule_dbg64_nopre.so timeout 200 python tools/timeline_w4.py 0 1 32 16384 0 3 > $O/s2_tl_d64_nopre.txt 2>&1; grep -v "^/opt" $O/s2_tl_d64_nopre.txt | head -24
