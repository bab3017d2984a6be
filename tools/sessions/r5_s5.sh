#!/bin/bash
# round 5, GPU session 5: whole GPU suite (tightened gradient bound, three backward modes); dK/dV timelines at D = 128 and D = 64 (finer
# stamps); ceiling control (vendor GEMM vs this build under one sampler) + both MFMA shapes of the bare probe; RMW next to MFMA;
# host cost per call; the bench line with the reference-harness block.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s5; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
for D in 128 64; do
  timeout 120 python tools/timeline_dkv4.py 1 8 32 32 2048 $D > $O/tl_dkv4_d$D.txt 2>&1; tail -5 $O/tl_dkv4_d$D.txt
done
timeout 120 python tools/timeline_dkv4.py 1 4 32 8 2048 128 > $O/tl_dkv4_c3.txt 2>&1; tail -5 $O/tl_dkv4_c3.txt
timeout 200 python tools/ceiling_control.py > $O/ceiling_control.txt 2>&1; cat $O/ceiling_control.txt
for m in 0 3; do timeout 60 build/probe_mfma_power 3 1 $m mode$m >> $O/probe_mfma.txt 2>&1; done; grep -E "^leg|TFLOP" $O/probe_mfma.txt | cut -c1-400
timeout 120 build/probe_mall > $O/probe_mall.txt 2>&1; tail -14 $O/probe_mall.txt
timeout 200 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; grep -E "host cost|queue kept|host time" $O/host_overhead.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt; tail -2 $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_s5/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
for r in d['extra']['ref_harness']['rows']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
PY
