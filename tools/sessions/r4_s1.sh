#!/bin/bash
# round 4, GPU session 1: power / clock telemetry under the kernels, the MFMA ceiling on random data, baseline bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s1; mkdir -p $O
LIB=aule-attention_amd/aule/lib/libaule.so
{
echo "### rocm-smi / amd-smi static"
timeout 60 rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | head -60
timeout 60 amd-smi static --limit 2>&1 | head -60
timeout 60 amd-smi metric -p -c 2>&1 | head -80
} > $O/smi_static.txt 2>&1
# background smi poll (cross-check of the sysfs reader), ~2 Hz
( for i in $(seq 1 40); do echo "t=$(date +%s.%N)"; timeout 20 rocm-smi --showpower --showclocks --json 2>/dev/null; sleep 0.3; done > $O/smi_poll.txt 2>&1 ) &
POLL=$!
PT_VERBOSE=1 timeout 120 build/power_trace $LIB fwd 4 32 32 4096 4096 128 bf16 1 3.0 1.0 c2_fwd_w4_random > $O/pt_c2_random.txt 2>&1
timeout 120 build/power_trace $LIB fwd 4 32 32 4096 4096 128 bf16 1 3.0 0.0 c2_fwd_w4_zeros > $O/pt_c2_zeros.txt 2>&1
AULE_HIP_FWD_KERNEL=ps timeout 120 build/power_trace $LIB fwd 4 32 32 4096 4096 128 bf16 1 3.0 1.0 c2_fwd_ps_random > $O/pt_c2_ps_random.txt 2>&1
timeout 120 build/power_trace $LIB bwd 4 32 8 2048 2048 128 bf16 1 2.5 1.0 c3_bwd_random > $O/pt_c3_bwd_random.txt 2>&1
timeout 120 build/power_trace $LIB bwd 4 32 8 2048 2048 128 bf16 1 2.5 0.0 c3_bwd_zeros > $O/pt_c3_bwd_zeros.txt 2>&1
kill $POLL 2>/dev/null
for spec in "1.0 0 mfma_random" "0.0 0 mfma_zeros" "1.0 1 mfma_random_4fma" "1.0 2 mfma_random_lds" "0.25 0 mfma_random_quarter_amp"; do
  set -- $spec
  timeout 120 build/probe_mfma_power 2.5 $1 $2 $3 > $O/probe_$3.txt 2>&1
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_bench_spawn.py -x -q > $O/pytest_dist.txt 2>&1
tail -3 $O/pytest_dist.txt
grep -h "mean after" $O/*.txt
