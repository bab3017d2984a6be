R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; V=$R/build/variants; cd $R
for n in nostore nostore_noslabread; do echo "== $n"; AULE_LIBRARY_PATH=$V/libaule_w4xd_$n.so timeout 200 python tools/timeline_w4.py 1 4 32 4096 0 3 2>&1 | grep -v "^/opt" | grep "epilogue\|prologue" | head -8; done
