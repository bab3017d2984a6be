R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for m in new old; do
  E=""; [ $m = old ] && E="AULE_HIP_BWD_DKV=old"
  env $E timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/bwpmc1_$m -- python $R/tools/bwd_one.py 2 16 16 4096 0 > $O/bwpmc1_$m.log 2>&1
  env $E timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/bwpmc2_$m -- python $R/tools/bwd_one.py 2 16 16 4096 0 > $O/bwpmc2_$m.log 2>&1
  echo "== $m"; python $R/tools/pmc_mean.py $O/bwpmc1_$m dk; python $R/tools/pmc_mean.py $O/bwpmc2_$m dk
done
rm -rf $O/bwpmc*/*/*kernel_trace.csv
