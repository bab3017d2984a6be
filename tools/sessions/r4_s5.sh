#!/bin/bash
# round 4, GPU session 5: whole GPU suite on the library without the stream kernel, host overhead of the autograd step, the fp32-atomic dQ probe
# re-priced with 256-key blocks, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s5; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 300 python tools/host_overhead.py > $O/host_overhead.txt 2>&1; head -3 $O/host_overhead.txt
for kb in 128 256; do for spin in 0 2000; do echo "## KB=$kb spin=$spin"; timeout 120 build/probe_atomic_kb$kb $spin; done; done > $O/probe_atomic.txt 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_s5/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
PY
cat $O/probe_atomic.txt
