#!/bin/bash
# round 4, GPU session 2: the embedded-request forward (literal scalar registers, fast first / prediag / diag bodies): parity, bit identity
# against the round-3 library, same-box A/B, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s2; mkdir -p $O
NEW=aule-attention_amd/aule/lib/libaule.so; OLD=build/variants/libaule_r3.so
{
for sh in "1 2 2 256 256 128 bf16 1" "1 2 2 300 300 128 bf16 1" "2 4 1 1024 1024 128 bf16 1" "1 3 3 1280 1280 128 bf16 1" "1 8 8 512 1024 128 bf16 2" \
          "1 2 2 200 333 128 bf16 0" "4 32 32 4096 4096 128 bf16 1" "4 32 8 2048 2048 128 bf16 1" "4 32 32 4096 4096 128 bf16 0" "1 32 1 16384 16384 64 fp16 0" \
          "2 8 8 1000 3000 128 bf16 2" "2 8 8 1111 1111 64 fp16 1" "1 8 8 8192 8192 128 bf16 1"; do
  for lib in $OLD $NEW; do echo "## $lib"; timeout 60 build/cbench $lib fwd $sh 20 10 10; done
done
} > $O/cbench_ab.txt 2>&1
timeout 900 python tools/w4_check.py check > $O/w4_check.txt 2>&1; tail -3 $O/w4_check.txt
timeout 300 python tools/timeline_w4.py 1 4 32 4096 0 3 > $O/timeline_c2.txt 2>&1
timeout 300 python tools/timeline_w4.py 1 4 32 2048 0 3 > $O/timeline_s2048.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_fwd_variants.py tests/test_gpu_rope.py tests/test_gpu_graph.py tests/test_gpu_bottom_right.py -x -q > $O/pytest_fwd.txt 2>&1; tail -3 $O/pytest_fwd.txt
grep -h "median" $O/cbench_ab.txt | paste - - | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,"old",$10,"new",$(NF/2+10)}' | head -20
python - <<'PY'
import re
t=open('gpurun_out/r4_s2/cbench_ab.txt').read().split('## ')
rows=[]
for blk in t[1:]:
    lib=blk.split('\n')[0]
    m=re.search(r'(fwd .*?): median ([\d.]+) us.*?([\d.]+) TF', blk)
    o=re.search(r'o: sum ([\-\d.e+]+) abs ([\-\d.e+]+)', blk)
    rows.append((lib, m.group(1) if m else '?', m.group(2) if m else '?', m.group(3) if m else '?', o.group(0) if o else '?'))
for a,b in zip(rows[0::2], rows[1::2]):
    print(a[1], '| old', a[2], 'us', a[3], 'TF | new', b[2], 'us', b[3], 'TF |', 'BIT-IDENTICAL' if a[4]==b[4] else 'DIFF '+a[4]+' vs '+b[4])
PY
