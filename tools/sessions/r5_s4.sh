#!/bin/bash
# round 5, GPU session 4: dQ = dS K kernel with two workgroups per CU; spill vs recompute per kernel on N(0,1) and on ZERO inputs (no
# power limit: is the dK/dV kernel's spill overhead cycles or joules?); non-temporal dS stores; HBM counters of both modes' dK/dV kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s4; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/one_bwd.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
B, Hq, Hkv, S, D = (int(x) for x in sys.argv[1:6]); n = int(sys.argv[6]); amp = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
dt = torch.bfloat16
q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt) * amp; k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt) * amp; v = torch.randn_like(k) * amp; do = torch.randn_like(q) * amp
sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, True, sc)
for _ in range(n): at.bwd_raw(q, k, v, out, do, lse, True, sc)
torch.cuda.synchronize()
PY
cat > /tmp/ks.py <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "fa_bwd" in n or "delta16" in n:
        a = float(r["AverageNs"]) / 1000; tot += a
        print("   %-26s calls %4s avg %8.1f us" % (n[n.find("fa_bwd"):][:26], r["Calls"], a))
print("   total %.1f us" % tot)
PY
run() {  # label, lib, mode, shape..., amp
  label=$1; lib=$2; mode=$3; shift 3
  if [ $lib = intree ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$lib.so; fi
  if [ $mode = recompute ]; then export AULE_HIP_BWD_MODE=recompute; else unset AULE_HIP_BWD_MODE; fi
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python /tmp/one_bwd.py "$@" > $O/kt.log 2>&1 < /dev/null )
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${label}_kernel_stats.csv
  rm -rf $O/kt
  echo "== $label: $*"; python /tmp/ks.py $O/${label}_kernel_stats.csv
}
for sh in "c3 4 32 8 2048 128 200" "c2 4 32 32 4096 128 60" "d64 8 32 32 2048 64 120"; do
  set -- $sh; tag=$1; shift
  run ${tag}_spill intree spill "$@" 1.0
  run ${tag}_spill_occ1 dqs_occ1 spill "$@" 1.0
  run ${tag}_spill_nt sp_nt spill "$@" 1.0
  run ${tag}_recompute intree recompute "$@" 1.0
  run ${tag}_spill_zeros intree spill "$@" 0.0
  run ${tag}_recompute_zeros intree recompute "$@" 0.0
done
unset AULE_LIBRARY_PATH
cd /tmp
for mode in spill recompute; do
  if [ $mode = recompute ]; then export AULE_HIP_BWD_MODE=recompute; else unset AULE_HIP_BWD_MODE; fi
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$mode/pmc_fetch -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$mode/pmc_write -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_write.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/$mode/pmc_sq2 -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_sq2.log 2>&1
  echo "===== PMC $mode"; (cd $R; python tools/summarize_prof.py $O/$mode | grep -v "w4_kernel" | cut -c1-160)
done
