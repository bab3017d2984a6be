#!/usr/bin/env python3
"""Mean of every counter per kernel in rocprofv3 counter_collection CSVs under the given directories, with the kernel's
mean duration from the same rows:  python tools/pmc_mean.py DIR [substring of the kernel name]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: [0.0, 0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want not in r["Kernel_Name"]:
            continue
        a = acc[(r["Kernel_Name"][:70], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (k, c), (s, n, t) in sorted(acc.items()):
    print(f"{k:70s} {c:28s} mean {s / n:16.1f}  n={n}  kernel {t / n:9.1f} us")
