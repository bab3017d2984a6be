#!/bin/bash
# round 4, GPU session 4: small grids on the one-wave-per-SIMD kernel (route 7 = key-range pieces + merge), the two-waves-per-SIMD stream retired
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s4; mkdir -p $O
NEW=aule-attention_amd/aule/lib/libaule.so; OLD=build/variants/libaule_r3.so
timeout 1200 python -m pytest tests/test_gpu_splitkv.py tests/test_gpu_graph.py tests/test_gpu_fwd_variants.py tests/test_gpu_fwd.py -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
{
for sh in "1 8 8 8192 8192 128 bf16 1" "1 8 8 4096 4096 128 bf16 1" "1 32 8 2048 2048 128 bf16 1" "1 32 32 2048 2048 64 fp16 1" "1 16 16 4096 4096 128 bf16 1" \
          "1 8 8 4096 4096 128 bf16 0" "1 8 8 2048 2048 128 bf16 0" "1 8 8 8192 8192 32 bf16 1" "4 32 32 4096 4096 128 bf16 1"; do
  for lib in $OLD $NEW; do echo "## $lib"; timeout 60 build/cbench $lib fwd $sh 20 10 10; done
  echo "## nosplit"; AULE_HIP_FWD_SPLIT=0 timeout 60 build/cbench $NEW fwd $sh 20 10 10
done
} > $O/cbench_ab.txt 2>&1
python - <<'PY'
import re
t=open('gpurun_out/r4_s4/cbench_ab.txt').read().split('## ')
rows=[]
for blk in t[1:]:
    lib=blk.split('\n')[0]
    m=re.search(r'(fwd .*?): median ([\d.]+) us.*?([\d.]+) TF', blk)
    rows.append((lib, m.group(1) if m else '?', m.group(2) if m else '?', m.group(3) if m else '?'))
for a,b,c in zip(rows[0::3], rows[1::3], rows[2::3]):
    print('%-44s r3 %8s us %7s TF | new %8s us %7s TF | new, no key split %8s us %7s TF' % (a[1], a[2], a[3], b[2], b[3], c[2], c[3]))
PY
