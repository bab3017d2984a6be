# tools/measure_all.sh -- one session on one box: smoke, the default bench line, every other config, the forward A/B against
# the predecessors, backward alone, decode / paged decode.  Run via gpurun; raw output -> profiles/r<N>_measure_all.txt.
set -u
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -1 gpurun_out/bench_final.json | python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r['extra']; f=r['roofline']
print('C2 driver-style (5+20): value %.1f TF  frac %.4f  steady_state %.1f TF (frac %.4f)  launch median/min/max %.1f/%.1f/%.1f us' % (r['value'], f['frac'], r['steady_state']['value'], f['frac_steady'], f['kernel_ms_median']*1e3, f['kernel_ms_min']*1e3, f['kernel_ms_max']*1e3))
print('C3 fwd+bwd %.1f TF (%.1f us; fwd %.1f us = %.0f TF, bwd %.1f us = %.0f TF)   C5 fwd %.1f TF (%.1f us)   cpu %.4f TF' % (e['c3_fwd_bwd_tflops'], e['c3_ms_per_step']*1e3, e['c3_fwd_ms']*1e3, e['c3_fwd_tflops'], e['c3_bwd_ms']*1e3, e['c3_bwd_tflops'], e['c5_fwd_tflops'], e['c5_ms_per_step']*1e3, r['cpu_baseline']['value']))"
for c in c4 c5 c5b c5c; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['workload'], 'ms', round(r['ms_per_step'],4), 'TF', round(r['value'],2), r['roofline']['bound'], round(r['roofline']['achieved'],1), r['roofline']['unit'], round(r['roofline']['frac'],4), 'steady', round(r['steady_state']['value'],2))"; done
python bench.py --config c3 --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['config']['workload'], 'ms', round(r['ms_per_step'],4), 'TF', round(r['value'],2), round(r['roofline']['frac'],4))"
echo "--- forward A/B: ping-pong kernel (AULE_HIP_FWD_KERNEL=pp) vs one wave per SIMD (default)"
AULE_HIP_FWD_KERNEL=pp python tools/fwd_check.py bench pp 2>&1 | grep -v amdgpu
python tools/fwd_check.py bench w4 2>&1 | grep -v amdgpu
python tools/w4_d64_check.py 2>&1 | grep "TF"
AULE_HIP_FWD_KERNEL=pp python tools/w4_d64_check.py 2>&1 | grep "TF"
echo "--- backward alone"
python tools/bwd_ab.py 2>&1 | grep -v amdgpu
echo "--- decode / paged decode"
python tools/bench_paged.py 2>&1 | tail -6
python tools/bench_decode_ws.py 2>&1 | tail -6
echo "--- RoPE + attention: fused query rotation vs two passes (tools/rope_ab.py)"
python tools/rope_ab.py 2>&1 | grep -v amdgpu
echo "--- small grids: route 7 on (default) / off (AULE_HIP_FWD_SPLIT=0)"
for sh in "1 8 8 8192 8192 128 bf16 1" "1 8 8 4096 4096 128 bf16 1" "1 16 16 4096 4096 128 bf16 1"; do build/cbench aule-attention_amd/aule/lib/libaule.so fwd $sh 20 10 10 | head -1; AULE_HIP_FWD_SPLIT=0 build/cbench aule-attention_amd/aule/lib/libaule.so fwd $sh 20 10 10 | head -1; done
echo "--- fp32 kernels"
python tools/f32_bench.py 2>&1 | grep -v amdgpu
