#!/bin/bash
# round 4, GPU session 17: the fp32 kernels with their work items in rank order (every unit's heaviest block first) against the
# build before it; the fp32 parity files; the dQ rule at D = 64 on small grids; the README's sliding-window shapes on the ping-pong route
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_fwd.py tests/test_gpu_capi.py tests/test_gpu_window.py tests/test_gpu_bottom_right.py -x -q -k "fp32 or f32 or legacy or capi" > $O/pytest_f32.txt 2>&1; tail -3 $O/pytest_f32.txt
AULE_LIBRARY_PATH=$PWD/build/variants/libaule_prerank.so timeout 300 python tools/f32_bench.py > $O/f32_before.txt 2>&1
timeout 300 python tools/f32_bench.py > $O/f32_after.txt 2>&1
paste -d'|' <(cut -c1-95 $O/f32_before.txt) <(cut -c50-120 $O/f32_after.txt)
L=aule-attention_amd/aule/lib/libaule.so
{
for sh in "1 8 8 1024 1024 64 bf16 1" "1 8 8 2048 2048 64 bf16 1" "1 16 16 2048 2048 64 bf16 1" "1 8 8 4096 4096 64 bf16 1" "1 32 32 2048 2048 64 fp16 1" "1 8 8 2048 2048 64 bf16 0" \
          "1 8 8 2048 2048 128 bf16 1" "1 16 16 2048 2048 128 bf16 1" "1 8 8 4096 4096 128 bf16 1"; do
  for m in old new; do echo "## $m"; AULE_HIP_BWD_DQ=$m timeout 100 build/cbench $L bwd $sh 20 10 5; done
done
} > $O/dq_rule.txt 2>&1
grep median $O/dq_rule.txt | paste - - | awk '{print $2,$3,$4,$5,$6,$7,"dq old",$9,"us | dq new",$23,"us"}'
timeout 300 python tools/window_bench.py > $O/window.txt 2>&1; cat $O/window.txt
