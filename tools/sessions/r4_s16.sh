#!/bin/bash
# round 4, GPU session 16: the D = 64 instances of the one-wave-per-SIMD dQ kernel: guarded probe (checksums against the predecessor:
# dQ is expected bit-identical), the backward test files, then kernel against kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s16; mkdir -p $O
L=aule-attention_amd/aule/lib/libaule.so
{
for sh in "1 2 2 128 128 64 bf16 0" "1 2 2 256 256 64 bf16 1" "2 8 8 512 512 64 bf16 1" "1 4 2 300 300 64 fp16 1" "2 8 8 1111 1111 64 fp16 1" "1 4 4 333 777 64 bf16 2"; do
  for m in old new; do echo "## $m"; AULE_HIP_BWD_DQ=$m AULE_HIP_BWD_DKV=$m timeout 60 build/cbench $L bwd $sh 5 2 1 || echo "FAILED rc=$?"; done
done
} > $O/probe.txt 2>&1
grep -c FAILED $O/probe.txt
python - <<'PY'
import re
t=open('gpurun_out/r4_s16/probe.txt').read().split('## ')
def sums(b): return re.findall(r'(dq|dk|dv): sum ([\-\d.e+]+) abs ([\-\d.e+]+)', b)
for a,b in zip(t[1::2], t[2::2]):
    sa, sb = sums(a), sums(b)
    print(a.split('\n')[1][:48] if len(a.split('\n'))>1 else '?', 'dq bit-identical' if sa[:1]==sb[:1] and sa else 'dq DIFF', ' old', sa, ' new', sb)
PY
if grep -q FAILED $O/probe.txt; then echo "probe failed; stopping"; exit 0; fi
timeout 1500 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_bottom_right.py tests/test_gpu_autograd.py -x -q > $O/pytest_bwd.txt 2>&1; tail -5 $O/pytest_bwd.txt
{
for sh in "8 32 32 2048 2048 64 bf16 1" "4 32 32 4096 4096 64 bf16 1" "8 32 8 2048 2048 64 bf16 1" "4 32 32 2048 2048 64 fp16 0" "1 32 1 16384 16384 64 fp16 1" "2 16 16 8192 8192 64 bf16 1" "16 16 16 1024 1024 64 bf16 1" "1 8 8 2048 2048 64 bf16 1"; do
  for m in old new; do echo "## $m"; AULE_HIP_BWD_DQ=$m timeout 100 build/cbench $L bwd $sh 20 10 5; done
  echo "## default"; timeout 100 build/cbench $L bwd $sh 20 10 5
done
} > $O/cbench_bwd_d64.txt 2>&1
grep "median" $O/cbench_bwd_d64.txt | paste - - - | cut -c1-330
