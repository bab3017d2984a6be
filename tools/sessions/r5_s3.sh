#!/bin/bash
# round 5, GPU session 3: (1) tools/probe_mall.hip: does the Infinity Cache keep written data for the next kernel; plain RMW throughput.
# (2) the dQ = dS K kernel's timing variants at C3 (stream alone / sequential layout / K side alone / ring depth).  (3) PMC passes of
# the C3 backward (5-matmul mode).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s3; mkdir -p $O
export TMPDIR=/tmp
timeout 120 build/probe_mall > $O/probe_mall.txt 2>&1; cat $O/probe_mall.txt
cat > /tmp/one_bwd.py <<'PY'
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "aule-attention_amd"))
from aule import _torch as at
B, Hq, Hkv, S, D = (int(x) for x in sys.argv[1:6]); n = int(sys.argv[6])
dt = torch.bfloat16
q = torch.randn(B, Hq, S, D, device="cuda", dtype=dt); k = torch.randn(B, Hkv, S, D, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
sc = 1 / math.sqrt(D); out, lse = at.fwd_raw(q, k, v, True, sc)
for _ in range(n): at.bwd_raw(q, k, v, out, do, lse, True, sc)
torch.cuda.synchronize()
PY
cat > /tmp/ks.py <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "fa_bwd" in n or "delta16" in n:
        print("   %-28s calls %4s avg %8.1f us" % (n.split("::")[-1].split("(")[0].split("<")[0] if False else n[n.find("fa_bwd"):][:28], r["Calls"], float(r["AverageNs"]) / 1000))
PY
for v in intree dqs_nomfma dqs_seq dqs_seq_nomfma dqs_nods dqs_a3 dqs_a2; do
  if [ $v = intree ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$R/build/variants/libaule_$v.so; fi
  for shape in "4 32 8 2048 128 40" "4 32 32 4096 128 20"; do
    tag=$(echo $shape | tr ' ' '_')
    ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python /tmp/one_bwd.py $shape > $O/kt.log 2>&1 < /dev/null )
    f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${v}_${tag}_kernel_stats.csv
    rm -rf $O/kt
    echo "== $v  $shape"; python /tmp/ks.py $O/${v}_${tag}_kernel_stats.csv
  done
done
unset AULE_LIBRARY_PATH
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/pmc_sq2 -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- python /tmp/one_bwd.py 4 32 8 2048 128 12 > $O/pmc_write.log 2>&1
cd $R; python tools/summarize_prof.py $O > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt | cut -c1-220
