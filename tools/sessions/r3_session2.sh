#!/bin/bash
# GPU session: round order of the causal part lists (parity, same-box A/B, HBM reads), the pre-scaled-Q form of the D = 64
# streams (parity, same-box A/B, cycle timelines), what SQ_INSTS_VALU counts.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
V=$R/build/variants
timeout 600 python tools/w4_check.py check quick > $O/s2_check.log 2>&1; tail -3 $O/s2_check.log
timeout 300 python tools/w4_d64_check.py > $O/s2_d64_pre.log 2>&1; cat $O/s2_d64_pre.log
AULE_LIBRARY_PATH=$V/libaule_nopre.so timeout 300 python tools/w4_d64_check.py > $O/s2_d64_nopre.log 2>&1; cat $O/s2_d64_nopre.log
timeout 300 python tools/w4_d64_check.py 2>&1 | grep "TF" > $O/s2_d64_pre2.log; cat $O/s2_d64_pre2.log
for i in 1 2; do
  AULE_HIP_W4_ORDER=pairs timeout 300 python tools/w4_check.py bench pairs$i > $O/s2_bench_pairs$i.log 2>&1; grep -h "bench\|TF" $O/s2_bench_pairs$i.log
  timeout 300 python tools/w4_check.py bench rounds$i > $O/s2_bench_rounds$i.log 2>&1; grep -h "bench\|TF" $O/s2_bench_rounds$i.log
done
W4_TL_D=64 W4_TL_HKV=1 AULE_LIBRARY_PATH=$V/libaule_dbg64.so timeout 200 python tools/timeline_w4.py 0 1 32 16384 0 > $O/s2_tl_d64_pre.txt 2>&1; grep -m3 "plain\|prologue" $O/s2_tl_d64_pre.txt
W4_TL_D=64 W4_TL_HKV=1 AULE_LIBRARY_PATH=$V/libaule_dbg64_nopre.so timeout 200 python tools/timeline_w4.py 0 1 32 16384 0 > $O/s2_tl_d64_nopre.txt 2>&1; grep -m3 "plain\|prologue" $O/s2_tl_d64_nopre.txt
AULE_LIBRARY_PATH=$V/libaule_dbg.so timeout 200 python tools/timeline_w4.py 1 4 32 4096 0 3 > $O/s2_tl_c2.txt 2>&1; grep -m2 "plain" $O/s2_tl_c2.txt
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-extra"
for ord in pairs rounds; do
  for pass in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    n=$(echo $pass | tr ' ' '_')
    AULE_HIP_W4_ORDER=$ord timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $O/s2_pmc_${ord}_$n -- python $R/bench.py $ARGS > $O/s2_pmc_${ord}_$n.log 2>&1
    echo "== $ord $pass"; python $R/tools/pmc_mean.py $O/s2_pmc_${ord}_$n fa_fwd_w4
  done
done
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU -d $O/s2_pmc_probe -- $R/build/probes/probe_trread > $O/s2_pmc_probe.log 2>&1
echo "== probe"; python $R/tools/pmc_mean.py $O/s2_pmc_probe | head -8
rm -rf $O/s2_pmc_*/*/*kernel_trace.csv
