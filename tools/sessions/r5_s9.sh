#!/bin/bash
# round 5, GPU session 9: kernel-only times (rocprofv3 kernel trace) of the fp32 legacy benchmark shape, split on / off -- how much of the
# wall time of a 30 us call is host issue?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s9; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/f32_legacy.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
B, H, S, D = 4, 8, 512, 64
q, k, v, do = (torch.randn(B, H, S, D, device="cuda") for _ in range(4))
sc = 1 / math.sqrt(D)
for causal in (False, True):
    out, lse = at.fwd_raw(q, k, v, causal, sc)
    for _ in range(200): at.fwd_raw(q, k, v, causal, sc)
    for _ in range(100): at.bwd_raw(q, k, v, out, do, lse, causal, sc)
torch.cuda.synchronize()
PY
for sp in 1 0; do
  ( cd /tmp; AULE_HIP_F32_SPLIT=$sp timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$sp -- python /tmp/f32_legacy.py > $O/kt$sp.log 2>&1 )
  f=$(find $O/kt$sp -name "*kernel_stats.csv" | head -1); cp $f $O/f32_legacy_split${sp}_kernel_stats.csv; rm -rf $O/kt$sp
  echo "== AULE_HIP_F32_SPLIT=$sp"; python - $O/f32_legacy_split${sp}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "f32" in n or "delta" in n:
        i = n.find("fa_"); print("   %-44s calls %4s avg %7.1f us  min %7.1f" % (n[i:][:44], r["Calls"], float(r["AverageNs"]) / 1000, float(r["MinNs"]) / 1000))
PY
done
