#!/bin/bash
# tools/profile.sh LABEL [bench args...]  -- run on the GPU box (via gpurun).
# Collects, for `python bench.py <args>`:
#   1. rocprofv3 --kernel-trace --stats (CSV)           -> gpurun_out/prof_LABEL/kt
#   2. PMC passes, each in its own run (never combined with trace domains other than
#      kernel-trace): SQ pass, FETCH_SIZE pass, WRITE_SIZE pass          -> .../pmc_*
# and prints compact summaries (tools/summarize_prof.py) into gpurun_out/prof_LABEL/summary.txt
set -u
LABEL=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$LABEL
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 50 --warmup 10 --no-cpu-baseline --no-extra $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py $ARGS > $OUT/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -- python $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -- python $R/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- python $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $R
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
