#!/bin/bash
# round 6, GPU session 3: the dK/dV stream block-major / head-minor (the GQA group's heads interleaved: an XCD's work items walk the same Q / dO rows
# together).  Parity in all modes, same-box A/B against the round-5 order (build/variants/libaule_r5dkv.so), FETCH_SIZE / WRITE_SIZE of the C3 step.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s3; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bottom_right.py tests/test_gpu_window.py -x -q -m gpu ) > $O/pytest_bwd.txt 2>&1; tail -4 $O/pytest_bwd.txt
for rep in 1 2; do
  for lib in r5dkv new; do
    if [ $lib = new ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
    echo "== $lib (rep $rep)"; timeout 300 python tools/bwd_ab.py 2>&1 | grep "bwd B"
  done
done > $O/bwd_ab.txt 2>&1
cat $O/bwd_ab.txt
export TMPDIR=/tmp; cd /tmp
for lib in r5dkv new; do
  if [ $lib = new ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
  ARGS="--config c3 --mode fwdbwd --steps 30 --warmup 5 --no-cpu-baseline --no-extra"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$lib/pmc_fetch -- python $R/bench.py $ARGS > $O/$lib.fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$lib/pmc_write -- python $R/bench.py $ARGS > $O/$lib.write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib/kt -- python $R/bench.py $ARGS > $O/$lib.kt.log 2>&1
  echo "===== $lib"; python $R/tools/summarize_prof.py $O/$lib
done > $O/c3_summary.txt 2>&1
cd $R; rm -rf $O/*/pmc_fetch $O/*/pmc_write $O/*/kt
cat $O/c3_summary.txt | cut -c1-170
