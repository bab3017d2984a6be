#!/bin/bash
# tools/pmc_ab.sh LABEL dtype B Hq Hkv S D causal -- SQ counters of one forward shape for the persistent tile stream and
# its predecessor (AULE_HIP_FWD_KERNEL=pp), two PMC passes each (no trace domain besides kernel-trace).  Run via gpurun.
set -u
LABEL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$LABEL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for K in pp ps; do
  if [ $K = pp ]; then export AULE_HIP_FWD_KERNEL=pp; else unset AULE_HIP_FWD_KERNEL; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/$K/pmc_sq -- python $R/tools/fwd_check.py one "$@" 12 > $OUT/$K.sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT/$K/pmc_sq2 -- python $R/tools/fwd_check.py one "$@" 12 > $OUT/$K.sq2.log 2>&1
  echo "===== $K"; python $R/tools/summarize_prof.py $OUT/$K | grep -v "^== kernel"
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
