#!/bin/bash
# round 6, GPU session 21 (final tree): smoke, the WHOLE GPU suite, the driver's bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s21; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt; tail -3 $O/bench_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_s21/bench_line.json') if l.startswith('{')][-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('frac' in k or 'window' in k)})
PY
