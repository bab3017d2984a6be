#!/bin/bash
# round 6, GPU session 1: what the box's amd-smi reports (limiter naming), the new full-size gradient tests + the `long` fuzz kind, and the
# re-instrumented bench line (hwmon sampler per leg, per-step medians, in-step backward events) with every step's time dumped
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_s1; mkdir -p $O
( timeout 60 amd-smi metric --help; echo ----; timeout 60 amd-smi metric -p -c --json; echo ----; timeout 60 amd-smi metric -v --json; echo ---; timeout 60 amd-smi metric -v ) > $O/amdsmi.txt 2>&1
ls /sys/class/drm/ > $O/sysfs.txt 2>&1; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h; cat $h/power1_label $h/power1_cap $h/freq1_input 2>&1; done >> $O/sysfs.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_bwd.py -q -x -m gpu -k "sizes_the_bench_times" -s > $O/pytest_fullsize.txt 2>&1; tail -3 $O/pytest_fullsize.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu -k "long" -s > $O/pytest_fuzz_long.txt 2>&1; tail -3 $O/pytest_fuzz_long.txt
AULE_BENCH_DUMP_STEPS=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt; tail -3 $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_s1/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1), d.get('telemetry_source'))
for k,v in d['legs'].items():
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_mean','ms_median','ms_min','ms_max','power_w','sclk_mhz','sclk_mhz_min','samples','steps_over_1p15_median')})
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('bwd_frac' in k)})
PY
