#!/bin/bash
# round 5, GPU session 19: the final tree's backward on the metric's shapes (no regression from the window paths) + the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s19; mkdir -p $O
timeout 300 python tools/bwd_ab.py > $O/bwd_ab.txt 2>&1; grep bwd $O/bwd_ab.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_s19/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('bwd' in k or 'fwd_bwd_tflops' in k)})
PY
