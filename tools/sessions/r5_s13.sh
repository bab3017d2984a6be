#!/bin/bash
# round 5, GPU session 13: the final tree -- whole GPU suite, smoke(), the split fuzz again (negative scale + rotation), the driver's bench line
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s13; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q --maxfail=8 > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 600 python tools/fuzz_parity.py split 60 5 > $O/fuzz_split.txt 2>&1; tail -2 $O/fuzz_split.txt
timeout 900 python tools/fuzz_parity.py 300 13 > $O/fuzz_default_13.txt 2>&1; tail -3 $O/fuzz_default_13.txt
AULE_HIP_BWD_MODE=spill AULE_HIP_BWD_DKV=new timeout 900 python tools/fuzz_parity.py 300 11 > $O/fuzz_spill_11.txt 2>&1; tail -3 $O/fuzz_spill_11.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_s13/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('achieved','frac','frac_steady','frac_zero_inputs','kernel_ms')})
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
for r in d['extra']['ref_harness']['rows']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('shape','aule_ms','aule_tflops','sdpa_tflops','speedup_vs_sdpa')})
print(json.dumps(d['roofline'].get('mfma_only_random_frac_of_peak'))[:300])
PY
