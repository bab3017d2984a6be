set -u
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -1 gpurun_out/bench_final.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('C2', r['value'], r['roofline']['frac'], 'c3', r['extra']['c3_fwd_bwd_tflops'], 'cpu', r['cpu_baseline']['value'])"
for c in c4 c5 c5b c5c; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['workload'], 'ms', round(r['ms_per_step'],4), 'TF', round(r['value'],2), r['roofline']['bound'], round(r['roofline']['achieved'],1), r['roofline']['unit'], round(r['roofline']['frac'],4))"; done
python tools/iw_check.py bench | tail -3
python tools/bench_paged.py | tail -5
