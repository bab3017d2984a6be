R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; V=$R/build/variants; cd $R
timeout 900 python tools/w4_check.py check > $O/s4_check.log 2>&1; grep -c "^ok" $O/s4_check.log; grep "FAIL\|ALL OK\|SOME" $O/s4_check.log
timeout 300 python tools/w4_d64_check.py > $O/s4_d64.log 2>&1; grep "FAIL\|D64\|TF" $O/s4_d64.log
timeout 300 python tools/w4_check.py bench new > $O/s4_bench.log 2>&1; grep "TF" $O/s4_bench.log
AULE_LIBRARY_PATH=$V/libaule_dbg.so timeout 200 python tools/timeline_w4.py 1 4 32 4096 0 3 > $O/s4_tl_c2.txt 2>&1; grep -v "^/opt" $O/s4_tl_c2.txt | head -12
AULE_LIBRARY_PATH=$V/libaule_dbg.so timeout 200 python tools/timeline_w4.py 0 4 32 4096 0 3 > $O/s4_tl_c2nc.txt 2>&1; grep "plain" $O/s4_tl_c2nc.txt | head -3
