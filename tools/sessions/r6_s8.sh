#!/bin/bash
# round 6, GPU session 8 (second working session of the round): the tree after the D = 64 two-key-block dK/dV instance: the driver's bench command,
# then the rocprofv3 records (kernel trace + SQ / FETCH / WRITE PMC passes, tools/profile.sh) of C2 forward, C2 / C3 / D = 64 fwd+bwd.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s8; mkdir -p $O
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt; tail -3 $O/bench_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_s8/bench_line.json') if l.startswith('{')][-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('frac' in k)})
PY
bash tools/profile.sh r6_c2 --config c2 > $O/profile_c2.txt 2>&1
bash tools/profile.sh r6_c2fb --config c2 --mode fwdbwd > $O/profile_c2fb.txt 2>&1
bash tools/profile.sh r6_c3fb --config c3 --mode fwdbwd > $O/profile_c3fb.txt 2>&1
P=$O/d64; mkdir -p $P; cd /tmp; export TMPDIR=/tmp
S=$R/tools/d64_step.py
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python $S > $P/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/pmc_sq -- python $S > $P/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $P/pmc_sq2 -- python $S > $P/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc_fetch -- python $S > $P/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc_write -- python $S > $P/pmc_write.log 2>&1
cd $R; python tools/summarize_prof.py $P > $O/profile_d64.txt 2>&1
for f in c2 c2fb c3fb; do cp gpurun_out/prof_r6_$f/summary.txt $O/summary_$f.txt; cp gpurun_out/prof_r6_$f/kt/*/*kernel_stats.csv $O/kernel_stats_$f.csv 2>/dev/null; done
cp $P/kt/*/*kernel_stats.csv $O/kernel_stats_d64.csv 2>/dev/null
rm -rf gpurun_out/prof_r6_* $P
head -12 $O/profile_d64.txt | cut -c1-160; head -8 $O/summary_c2fb.txt | cut -c1-160
python tools/window_bench.py 2>&1 | grep -v amdgpu > $O/window_bench.txt; cat $O/window_bench.txt | cut -c1-200
