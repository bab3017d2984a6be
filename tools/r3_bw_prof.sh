R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for m in new old; do
  E=""; [ $m = old ] && E="AULE_HIP_BWD_DKV=old"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bwprof_$m -- python $R/tools/bwd_ab.py > $O/bwprof_$m.log 2>&1
  echo "== $m"; f=$(ls $O/bwprof_$m/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "bwd" in n or "dkv" in n:
        print(f"  {n[:95]:95s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
rm -rf $O/bwprof_*/*/*kernel_trace.csv
