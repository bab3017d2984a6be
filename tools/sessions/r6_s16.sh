#!/bin/bash
# round 6, GPU session 16: the tree as it stands -- the WHOLE GPU suite, smoke, the driver's bench command, tools/measure_all.sh, fuzz (general, long, window).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s16; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench_err.txt; tail -3 $O/bench_err.txt
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6_s16/bench_line.json') if l.startswith('{')][-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('frac' in k or 'window' in k)})
PY
( timeout 1500 bash tools/measure_all.sh ) > $O/measure_all.txt 2>&1; tail -60 $O/measure_all.txt | cut -c1-220
( timeout 900 python tools/fuzz_parity.py 300 9; timeout 900 python tools/fuzz_parity.py long 24 2; timeout 600 python tools/fuzz_parity.py window 200 5 ) 2>&1 | grep -v amdgpu > $O/fuzz.txt; tail -6 $O/fuzz.txt | cut -c1-300
