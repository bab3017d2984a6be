R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for n in ${VARS:-quad_count quad_cost pipe_count pipe_cost pipe_cost22}; do echo "== $n"; for i in 1 2; do AULE_LIBRARY_PATH=$R/build/variants/libaule_w4xd_$n.so timeout 300 python tools/timeline_w4.py 0 4 32 4096 0 2>&1 | grep "plain" | head -1; done; done
