#!/bin/bash
# round 5, GPU session 15: per-kernel times of a windowed backward (B4 H32 S4096 D128 W256), new pair vs predecessors
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s15; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/win_bwd.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
B, H, S, D, W = 4, 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
q, k, v, do = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(4))
sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, True, sc, window=W)
for _ in range(60): at.bwd_raw(q, k, v, out, do, lse, True, sc, window=W)
torch.cuda.synchronize()
PY
for leg in new old; do
  for sh in "4096 128 256" "8192 128 1024"; do
    ( cd /tmp; AULE_HIP_BWD_DKV=$leg AULE_HIP_BWD_DQ=$leg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python /tmp/win_bwd.py $sh > $O/kt.log 2>&1 )
    f=$(find $O/kt -name "*kernel_stats.csv" | head -1); tag=$(echo $sh | tr ' ' '_'); cp $f $O/${leg}_${tag}_kernel_stats.csv; rm -rf $O/kt
    echo "== $leg  S D W = $sh"; python - $O/${leg}_${tag}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "fa_bwd" in n:
        i = n.find("fa_bwd"); print("   %-40s calls %4s avg %8.1f us" % (n[i:][:40], r["Calls"], float(r["AverageNs"]) / 1000))
PY
  done
done
