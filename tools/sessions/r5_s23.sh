#!/bin/bash
# round 5, GPU session 23: the whole GPU suite + smoke + the driver's bench command on the tree with the trimmed dK/dV loop
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s23; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_s23/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('bwd' in k or 'fwd_bwd_tflops' in k)})
PY
