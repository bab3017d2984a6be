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

# machine-readable HBM traffic for bench.py (KB counters; gfx950: FETCH_SIZE counts 128-B requests as 64 B)
import json
traffic = {}
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(sub, "*counter_collection.csv"):
        tot, n = 0.0, 0
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "fa_fwd" in row.get("Kernel_Name", "") and row.get("Counter_Name") == key:
                    tot += float(row["Counter_Value"]); n += 1
        if n:
            traffic[key] = tot / n
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    b = (2.0 * traffic["FETCH_SIZE"] + traffic["WRITE_SIZE"]) * 1024.0
    print("== HBM traffic per fwd launch: read %.1f MB (2 x FETCH_SIZE) + write %.1f MB = %.1f MB" % (
        2 * traffic["FETCH_SIZE"] * 1024 / 1e6, traffic["WRITE_SIZE"] * 1024 / 1e6, b / 1e6))
    with open(os.path.join(root, "hbm_traffic.json"), "w") as fh:
        json.dump({"c2_fwd": {"bytes_per_launch": b, "fetch_size_kb": traffic["FETCH_SIZE"],
                              "write_size_kb": traffic["WRITE_SIZE"]}}, fh)
