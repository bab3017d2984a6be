#!/bin/bash
# round 6, GPU session 4: the dK/dV stream in lock step (causal pairs: the long part first, walked downwards, the short part upwards; head-minor).
# Parity in all modes; same-box A/B vs the round-5 order; FETCH_SIZE / WRITE_SIZE at C3 and C2; then the 5-matmul mode re-measured on top of it
# (AULE_HIP_BWD_MODE=spill; dS stores plain / non-temporal at the start / end of phase 2) at C3, C2, D = 64.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s4; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bottom_right.py tests/test_gpu_window.py -x -q -m gpu ) > $O/pytest_bwd.txt 2>&1; tail -4 $O/pytest_bwd.txt
for rep in 1 2; do
  for lib in r5dkv new; do
    if [ $lib = new ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
    echo "== $lib recompute (rep $rep)"; AULE_HIP_BWD_MODE=recompute timeout 300 python tools/bwd_ab.py 2>&1 | grep "bwd B"
  done
  for lib in new sp_start_1 sp_end_1; do
    if [ $lib = new ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
    echo "== $lib spill (rep $rep)"; AULE_HIP_BWD_MODE=spill timeout 300 python tools/bwd_ab.py 2>&1 | grep "bwd B"
  done
done > $O/bwd_ab.txt 2>&1
cat $O/bwd_ab.txt
unset AULE_LIBRARY_PATH
export TMPDIR=/tmp; cd /tmp
for cfg in c3 c2; do
for lib in r5dkv new; do
  if [ $lib = new ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
  ARGS="--config $cfg --mode fwdbwd --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$cfg$lib/pmc_fetch -- python $R/bench.py $ARGS > $O/$cfg$lib.fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$cfg$lib/pmc_write -- python $R/bench.py $ARGS > $O/$cfg$lib.write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$cfg$lib/kt -- python $R/bench.py $ARGS > $O/$cfg$lib.kt.log 2>&1
  echo "===== $cfg $lib"; python $R/tools/summarize_prof.py $O/$cfg$lib
done
done > $O/summary.txt 2>&1
cd $R; rm -rf $O/*/pmc_fetch $O/*/pmc_write $O/*/kt $O/*.log
grep -E "=====|fa_bwd|FETCH|WRITE" $O/summary.txt | cut -c1-150
