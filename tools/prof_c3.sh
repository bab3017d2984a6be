#!/bin/bash
# tools/prof_c3.sh -- rocprofv3 kernel trace of the C3 forward+backward step (bench.py --config c3 --mode fwdbwd); summary on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r3b_c3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --config c3 --mode fwdbwd --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $OUT/kt.log 2>&1 < /dev/null
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $OUT/kernel_stats.csv; cut -c1-220 $f | head -8; else echo "no kernel_stats.csv"; tail -5 $OUT/kt.log; fi
tail -1 $OUT/kt.log | cut -c1-300
