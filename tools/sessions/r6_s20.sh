#!/bin/bash
# round 6, GPU session 20: rocprofv3 records of a sliding-window training step (B4 H32 S8192 D128 bf16 causal, W 256 and W 1024: the window instances of the forward, the
# dQ and dK/dV kernels): kernel trace + SQ / FETCH / WRITE PMC passes, each in its own run (tools/profile.sh's passes on tools/window_step.py).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s20; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for W in 256 1024; do
  P=$O/w$W; mkdir -p $P; S="$R/tools/window_step.py $W"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python $S > $P/kt.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/pmc_sq -- python $S > $P/pmc_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $P/pmc_sq2 -- python $S > $P/pmc_sq2.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc_fetch -- python $S > $P/pmc_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc_write -- python $S > $P/pmc_write.log 2>&1
  ( cd $R; python tools/summarize_prof.py $P ) > $O/summary_w$W.txt 2>&1
  cp $P/kt/*/*kernel_stats.csv $O/kernel_stats_w$W.csv 2>/dev/null
  rm -rf $P
done
head -8 $O/summary_w256.txt | cut -c1-170
