#!/usr/bin/env python3
"""Summarises the legs of tools/power_trace.cpp / tools/probe_mfma_power.hip (gpurun_out/<session>/*.txt) into one table:
which card is ours (the one whose power moves), socket power, power limit, shader clock, launch time, TFLOP/s.
    python tools/summarize_power.py gpurun_out/r4_s1 > profiles/r4_power_trace.txt"""
import glob
import os
import re
import sys

PEAK = 2516.6


def legs(path):
    out = []
    for f in sorted(glob.glob(os.path.join(path, "*.txt"))):
        txt = open(f).read()
        m = re.search(r"^leg ([^/\n:]+): .*$", txt, re.M)
        if not m:
            continue
        cards = {}
        for c in re.finditer(r"(card\d+)\.(\w+) mean ([\d.e+]+) min ([\d.e+]+) max ([\d.e+]+)", m.group(0)):
            cards.setdefault(c.group(1), {})[c.group(2)] = tuple(float(c.group(i)) for i in (3, 4, 5))
        hdr = re.search(r"^== (.*)$", txt, re.M)
        mean = re.search(r"mean after 500 ms: ([\d.]+) us per launch = ([\d.]+) TFLOP/s", txt)
        out.append((m.group(1), hdr.group(1) if hdr else "", cards, mean, txt))
    return out


def main():
    path = sys.argv[1]
    L = legs(path)
    # our card: the one with the largest power swing across all legs
    swing = {}
    for _, _, cards, _, _ in L:
        for c, d in cards.items():
            p = d.get("power1_input", (0, 0, 0))[0]
            lo, hi = swing.get(c, (1e30, 0))
            swing[c] = (min(lo, p), max(hi, p))
    ours = max(swing, key=lambda c: (swing[c][1] - swing[c][0]) + swing[c][1])   # moves the most / draws the most
    print("# Socket power, power limit and shader clock WHILE the kernels run (hwmon power1_input / power1_cap / freq1_input of the")
    print("# device under test, sampled every 10 ms by tools/power_sampler.h; means over the samples after the first 500 ms of back-to-back")
    print("# launches).  The box shows the 8 cards of its node in sysfs; the device under test is the one whose power moves: %s." % ours)
    print("# Other cards of the node (other tenants): idle or busy independently, not shown.")
    print("#")
    print("# %-28s %9s %8s %8s %9s %9s %9s %7s" % ("leg", "us/launch", "TFLOP/s", "of peak", "power W", "(min-max)", "limit W", "sclk MHz"))
    for name, hdr, cards, mean, _ in L:
        d = cards[ours]
        p, s, cap = d["power1_input"], d["freq1_input"], d["power1_cap"]
        us, tf = (float(mean.group(1)), float(mean.group(2))) if mean else (float("nan"), float("nan"))
        print("  %-28s %9.1f %8.1f %8.3f %9.1f %4.0f-%-4.0f %9.0f %7.0f   (%.0f-%.0f)" % (
            name, us, tf, tf / PEAK, p[0] / 1e6, p[1] / 1e6, p[2] / 1e6, cap[0] / 1e6, s[0] / 1e6, s[1] / 1e6, s[2] / 1e6))
    print()
    for name, hdr, cards, mean, txt in L:
        print("## %s: %s" % (name, hdr))
        tr = re.search(r"  trace \(t_ms, (.*)\):\n((?:   .*\n)+)", txt)
        if tr:
            cols = tr.group(1).split(", ")
            ip, iq = cols.index(ours + ".power1_input"), cols.index(ours + ".freq1_input")
            print("   t_ms   power_W  sclk_MHz   (every 10th sample)")
            for row in tr.group(2).strip("\n").split("\n")[:40]:
                v = row.split()
                if len(v) > max(ip, iq) + 1:
                    print("   %7.0f %8.1f %8.0f" % (float(v[0]), float(v[1 + ip]) / 1e6, float(v[1 + iq]) / 1e6))
        lt = re.search(r"launch time per window.*\n((?:   .*\n)+)", txt)
        if lt:
            print("   launch windows (t_ms, us per launch, TFLOP/s): " + " | ".join(" ".join(r.split()) for r in lt.group(1).strip("\n").split("\n")[:10]))
        print()


if __name__ == "__main__":
    main()
