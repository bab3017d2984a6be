R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; V=$R/build/variants; cd $R
timeout 900 python tools/w4_check.py check > $O/s7_check.log 2>&1; grep -c "^ok" $O/s7_check.log; grep "FAIL\|ALL OK\|SOME" $O/s7_check.log
timeout 300 python tools/w4_d64_check.py > $O/s7_d64.log 2>&1; grep "FAIL\|D64\|TF" $O/s7_d64.log
for i in 1 2; do
echo "== old (quad order, DMA piece three gaps from the end)"; AULE_LIBRARY_PATH=$V/libaule_w4x_old.so timeout 300 python tools/w4_check.py bench old 2>&1 | grep TF
echo "== new (pipelined order, DMA piece in the last gap)"; timeout 300 python tools/w4_check.py bench new 2>&1 | grep TF
done
AULE_LIBRARY_PATH=$V/libaule_w4x_old.so timeout 300 python tools/w4_d64_check.py 2>&1 | grep "TF"
