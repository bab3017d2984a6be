#!/bin/bash
# round 4, GPU session 21: the round's final build -- whole GPU suite, the backward fuzz with both new kernels forced, the C2 forward profile
# (kernel trace + PMC passes), C2 / C3 forward+backward kernel traces, the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s21; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
AULE_HIP_BWD_DKV=new AULE_HIP_BWD_DQ=new timeout 600 python tools/fuzz_parity.py 250 7 > $O/fuzz_new.txt 2>&1; tail -3 $O/fuzz_new.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
bash tools/profile.sh r4_c2 --config c2 > $O/profile_c2.txt 2>&1
export TMPDIR=/tmp
for cfg in c2 c3; do
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -- python $R/bench.py --config $cfg --mode fwdbwd --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $O/kt_$cfg.log 2>&1 < /dev/null )
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${cfg}_fwdbwd_kernel_stats.csv
  rm -rf $O/kt_$cfg
done
tail -32 $O/profile_c2.txt | cut -c1-160
for cfg in c2 c3; do echo "== $cfg fwd+bwd"; cut -c1-200 $O/${cfg}_fwdbwd_kernel_stats.csv | head -5; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_s21/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1), 'roofline', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('achieved','frac','frac_steady','frac_zero_inputs')})
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
print('cpu_baseline', d.get('cpu_baseline'))
PY
