#!/bin/bash
# round 5, GPU session 2: the dS stores of the SPILL dK/dV kernel (placement x non-temporal) and the dS LDS-DMA source pattern of the
# dQ = dS K kernel, same box; per-kernel times from a kernel trace of the C3 / C2 / D64 backward per variant.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s2; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/one_bwd.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
def bwd(B, Hq, Hkv, S, causal, D=128, dt=torch.bfloat16, n=40):
    q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, causal, sc)
    for _ in range(n): at.bwd_raw(q, k, v, out, do, lse, causal, sc)
    torch.cuda.synchronize()
bwd(4, 32, 8, 2048, True); bwd(4, 32, 32, 4096, True, n=25); bwd(8, 32, 32, 2048, True, 64)
PY
for v in intree sp_start_1 sp_end_0 sp_end_1 sp_noquad; do
  if [ $v = intree ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$v.so; fi
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- python /tmp/one_bwd.py > $O/kt_$v.log 2>&1 < /dev/null )
  f=$(find $O/kt_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${v}_kernel_stats.csv
  rm -rf $O/kt_$v
  echo "== $v"; grep -E "fa_bwd|delta16" $O/${v}_kernel_stats.csv | sed -E 's/"void aule_hip::\(anonymous namespace\):://; s/\(aule_hip.*Params\)"//' | awk -F, '{printf "%-70s calls %s avg %.1f us\n", $1, $2, $4/1000}'
done
unset AULE_LIBRARY_PATH
timeout 300 python tools/bwd_ab.py > $O/bwd_ab_intree.txt 2>&1; cat $O/bwd_ab_intree.txt
