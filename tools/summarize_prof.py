#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a few lines per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


for f in find("kt", "*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print("  " + ",".join(x[:60] for x in row))

for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")[:70]
                c = row.get("Counter_Name")
                acc[k][c] += float(row.get("Counter_Value", 0))
                cnt[k][c] += 1
        print("== PMC (mean per dispatch):", sub)
        for k in acc:
            if "fa_" not in k:
                continue
            print("  ", k)
            for c in sorted(acc[k]):
                print("      %-32s %16.1f   (n=%d)" % (c, acc[k][c] / max(1, cnt[k][c]), cnt[k][c]))
