#!/bin/bash
# round 5, GPU session 1: first run of the 5-matmul backward (delta pass + dK/dV with dS spill + dQ = dS K): parity in the three
# backward modes, then same-box timing spill vs recompute, then evidence on the round-4 pair (PMC passes of dq4 / dkv4 at C2).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bottom_right.py -m gpu -x -q > $O/pytest_bwd.txt 2>&1; tail -5 $O/pytest_bwd.txt
for mode in spill recompute; do
  if [ $mode = recompute ]; then export AULE_HIP_BWD_MODE=recompute; else unset AULE_HIP_BWD_MODE; fi
  timeout 300 python tools/bwd_ab.py > $O/bwd_ab_$mode.txt 2>&1; echo "== $mode"; cat $O/bwd_ab_$mode.txt
done
unset AULE_HIP_BWD_MODE
AULE_HIP_DQS_REV=0 timeout 300 python tools/bwd_ab.py > $O/bwd_ab_spill_norev.txt 2>&1; echo "== spill, dqs grid forward"; cat $O/bwd_ab_spill_norev.txt
export TMPDIR=/tmp
for cfg in c3 c2; do
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -- python $R/bench.py --config $cfg --mode fwdbwd --steps 30 --warmup 10 --no-cpu-baseline --no-extra > $O/kt_$cfg.log 2>&1 < /dev/null )
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${cfg}_fwdbwd_kernel_stats.csv
  rm -rf $O/kt_$cfg
  echo "== $cfg fwd+bwd (spill)"; cut -c1-160 $O/${cfg}_fwdbwd_kernel_stats.csv | head -7
done
