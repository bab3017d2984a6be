#!/bin/bash
# round 4, GPU session 11: the last forward build (seam, K_3 in step 0, pointers in registers): whole GPU suite, bit identity, A/B, timeline, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s11; mkdir -p $O
NEW=aule-attention_amd/aule/lib/libaule.so; OLD=build/variants/libaule_r3.so
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
{
for sh in "1 2 2 256 256 128 bf16 1" "1 2 2 300 300 128 bf16 1" "2 4 1 1024 1024 128 bf16 1" "1 3 3 1280 1280 128 bf16 1" "1 8 8 512 1024 128 bf16 2" \
          "1 2 2 200 333 128 bf16 0" "4 32 32 4096 4096 128 bf16 1" "4 32 8 2048 2048 128 bf16 1" "4 32 32 4096 4096 128 bf16 0" "1 32 1 16384 16384 64 fp16 0" \
          "2 8 8 1111 1111 64 fp16 1" "16 16 16 1024 1024 128 bf16 1" "8 32 32 2048 2048 64 bf16 1" "8 32 32 8192 8192 128 bf16 1"; do
  for lib in $OLD $NEW; do echo "## $lib"; timeout 100 build/cbench $lib fwd $sh 20 10 10; done
done
} > $O/cbench_ab.txt 2>&1
python - <<'PY'
import re
t=open('gpurun_out/r4_s11/cbench_ab.txt').read().split('## ')
rows=[]
for blk in t[1:]:
    lib=blk.split('\n')[0]
    m=re.search(r'(fwd .*?): median ([\d.]+) us.*?([\d.]+) TF', blk)
    o=re.search(r'o: sum ([\-\d.e+]+) abs ([\-\d.e+]+)', blk)
    rows.append((lib, m.group(1) if m else '?', m.group(2) if m else '?', m.group(3) if m else '?', o.group(0) if o else '?'))
for a,b in zip(rows[0::2], rows[1::2]):
    print('%-46s r3 %8s us %7s TF | new %8s us %7s TF | %+5.1f %% | %s' % (a[1], a[2], a[3], b[2], b[3], (float(a[2])/float(b[2])-1)*100, 'bit-identical' if a[4]==b[4] else 'DIFF'))
PY
for lib in $OLD $NEW; do CB_AMP=0 timeout 60 build/cbench $lib fwd 4 32 32 4096 4096 128 bf16 1 30 20 10 | head -1 | sed "s|^|zeros $lib: |"; done
timeout 300 python tools/timeline_w4.py 1 4 32 4096 0 3 > $O/timeline_c2.txt 2>&1; grep workgroup $O/timeline_c2.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_s11/bench_line.json').read().strip().splitlines()[-1])
print('value',round(d['value'],1),'steady',round(d['steady_state']['value'],1))
print({k:round(v,3) for k,v in d['extra'].items() if isinstance(v,float) and ('tflops' in k or 'frac' in k)})
PY
