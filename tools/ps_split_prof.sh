#!/bin/bash
# GPU-side kernel durations of the causal split with 0 / 2 / 4 / 8 pieces at most (rocprofv3 kernel trace -> sqlite).
cd /tmp && export TMPDIR=/tmp
for n in 0 2 4 8; do
  rm -rf /tmp/psp_$n
  AULE_HIP_FWD_PSSPLIT=$n timeout 150 rocprofv3 --kernel-trace -d /tmp/psp_$n -o t -- python /root/repo/tools/ps_split_check.py 0 1 3 5 7 9 11 12 < /dev/null > /tmp/psp_$n.log 2>&1
  echo "== max pieces $n"; grep -E "causal:" /tmp/psp_$n.log
  python3 - /tmp/psp_$n/t_results.db <<'PY'
import sqlite3, sys, statistics
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like '%aule%' order by d.start").fetchall()
# a split launch = stream kernel + merge kernel: key both by the merge kernel's grid (Q blocks, heads) = the shape
g, pend = {}, None
for name, gx, gy, gz, st, en in rows:
    us = (en - st) / 1e3
    if "combine" in name:
        key = "split  nqb %-3d heads %-3d items %-4d" % (gy, gz, pend[0])
        g.setdefault(key, []).append((pend[1], us))
        pend = None
    elif "ELb0ELb0ELb1EEE" in name:
        pend = (gx // 512, us)
    else:
        g.setdefault("plain  items %-4d %s" % (gx // 512, "D64" if "Li64" in name else "D128"), []).append((us, 0.0))
for k, v in g.items():
    v = v[len(v) // 2:]
    a, b = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
    print("   %-40s stream %7.1f us + merge %5.1f us = %7.1f us" % (k, a, b, a + b))
PY
done
