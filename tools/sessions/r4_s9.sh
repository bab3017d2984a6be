#!/bin/bash
# round 4, GPU session 9: the round's profiles -- C2 forward (kernel trace + PMC passes), C2 and C3 forward+backward kernel traces, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s9; mkdir -p $O
bash tools/profile.sh r4_c2 --config c2 > $O/profile_c2.txt 2>&1
export TMPDIR=/tmp
for cfg in c2 c3; do
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -- python $R/bench.py --config $cfg --mode fwdbwd --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/kt_$cfg.log 2>&1 < /dev/null )
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${cfg}_fwdbwd_kernel_stats.csv
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
timeout 900 python -m pytest tests/test_gpu_fwd_variants.py -x -q > $O/pytest_variants.txt 2>&1; tail -2 $O/pytest_variants.txt
cat $O/profile_c2.txt | tail -40
for cfg in c2 c3; do echo "== $cfg fwd+bwd"; cut -c1-200 $O/${cfg}_fwdbwd_kernel_stats.csv | head -6; done
