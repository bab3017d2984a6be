#!/bin/bash
# round 5, GPU session 8: profiles of the round's build -- C2 forward (kernel trace + PMC passes), C2 / C3 forward+backward (kernel trace +
# PMC passes), the D = 64 training shape (kernel trace + PMC), power traces of the backward, the driver's bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s8; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile.sh r5_c2 --config c2 > $O/profile_c2.txt 2>&1; tail -40 $O/profile_c2.txt | cut -c1-170
bash tools/profile.sh r5_c2fb --config c2 --mode fwdbwd > $O/profile_c2fb.txt 2>&1
bash tools/profile.sh r5_c3fb --config c3 --mode fwdbwd > $O/profile_c3fb.txt 2>&1
cat > /tmp/d64_step.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
import aule
g = torch.Generator(device="cuda").manual_seed(99)
q, k, v = (torch.randn(8, 32, 2048, 64, device="cuda", dtype=torch.bfloat16, generator=g).requires_grad_(True) for _ in range(3))
d = torch.randn(8, 32, 2048, 64, device="cuda", dtype=torch.bfloat16, generator=g)
for _ in range(60):
    q.grad = k.grad = v.grad = None
    aule.flash_attention(q, k, v, causal=True).backward(d)
torch.cuda.synchronize()
PY
P=$O/d64; mkdir -p $P; cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python /tmp/d64_step.py > $P/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $P/pmc_sq -- python /tmp/d64_step.py > $P/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $P/pmc_sq2 -- python /tmp/d64_step.py > $P/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc_fetch -- python /tmp/d64_step.py > $P/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc_write -- python /tmp/d64_step.py > $P/pmc_write.log 2>&1
cd $R; python tools/summarize_prof.py $P > $O/profile_d64.txt 2>&1
L=$R/aule-attention_amd/aule/lib/libaule.so
{
for leg in "bwd 4 32 8 2048 2048 128 bf16 1 3 1 c3_bwd" "bwd 4 32 8 2048 2048 128 bf16 1 3 0 c3_bwd_zeros" "bwd 4 32 32 4096 4096 128 bf16 1 3 1 c2_bwd" "bwd 8 32 32 2048 2048 64 bf16 1 3 1 d64_bwd" "bwd 8 32 32 2048 2048 64 bf16 1 3 0 d64_bwd_zeros" "fwd 4 32 32 4096 4096 128 bf16 1 3 1 c2_fwd"; do
  timeout 60 build/power_trace $L $leg 2>&1 | grep -E "^leg|mean after|us per launch =" | cut -c1-900
done
} > $O/power_trace.txt 2>&1
python tools/summarize_power.py $O/power_trace.txt > $O/power_summary.txt 2>&1; cat $O/power_summary.txt | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_s8/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('achieved','frac','frac_steady','frac_zero_inputs','kernel_ms')})
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
for r in d['extra']['ref_harness']['rows']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('shape','aule_ms','aule_tflops','sdpa_tflops','speedup_vs_sdpa')})
PY
for f in profile_c2fb profile_c3fb profile_d64; do echo "===== $f"; grep -A9 -E "fa_bwd|fa_fwd_w4" $O/$f.txt | grep -E "aule_hip|GRBM|MFMA_BUSY|INSTS_MFMA|INSTS_VALU|INSTS_LDS|INSTS_SALU|FETCH|WRITE" | cut -c1-120; grep -E "Average|kernel_stats|us " $O/$f.txt | head -12 | cut -c1-200; done
