#!/usr/bin/env python3
"""Issue-sum model of the one-wave-per-SIMD kernels, from the COMPILED assembly (no GPU): a lone wave issues one instruction every
~4.6 cycles whatever its kind -- more for the kinds below -- and the period of a loop body is the SUM of those costs, not the
maximum over the pipes (profiles/r3_probe_fillers.txt for the prices, profiles/r3b_bwd_dkv4_steps.txt for the check: 1734 modelled
against 1730 measured for the dK/dV iteration, 2716 against 2854 for the forward's plain step).

    python tools/issue_sum.py <file.hip | file.s> [kernel-name-substring]

compiles a .hip with the product's flags (hipcc -S --cuda-device-only), finds the loops of every matching kernel (backward
branches) and prints, for each loop with MFMAs in it: instructions by class (compiler-generated ones outside the asm statements
counted separately), the modelled cycles per trip and per MFMA, and the matrix pipe's own time (32 cycles per MFMA).  A loop that
contains several variants of a body laid out one after the other (masked / tail forms) is priced as laid out: read the steady
loops (the smallest ones that hold the expected number of MFMAs)."""
import os
import re
import subprocess
import sys
import tempfile

COST = {"mfma": 8.0, "valu": 4.6, "exp": 8.6, "b128": 16.0, "tr": 8.0, "lds_other": 8.0, "dma": 26.4, "vmem": 12.0, "salu": 4.6,
        "wait": 4.6, "barrier": 4.6, "nop": 4.6}
# (a DMA piece = its s_add m0 (salu) + the request: 31 together)


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op == "v_exp_f32":
        return "exp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read_b128"):
        return "b128"
    if op.startswith("ds_read_b64_tr"):
        return "tr"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"):
        return "dma" if t.rstrip().endswith("lds") else "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return None


def kernels(text):
    cur, name, out = None, None, []
    for l in text.split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m and "kernel" in m.group(1):
            name, cur = m.group(1), []
        elif l.startswith(".Lfunc_end") and cur is not None:
            out.append((name, cur))
            cur = None
        elif cur is not None:
            cur.append(l)
    return out


def loops(body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    res = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            res.append((labels[m.group(1)], i))
    return sorted(set(res), key=lambda ab: ab[1] - ab[0])


def price(lines):
    inasm, n, gen = False, {}, {}
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        c = classify(t)
        if c is None:
            continue
        n[c] = n.get(c, 0) + 1
        if not inasm:
            gen[c] = gen.get(c, 0) + 1
    cyc = sum(COST[c] * k for c, k in n.items())
    return n, gen, cyc


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    if path.endswith(".hip"):
        out = os.path.join(tempfile.mkdtemp(), "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-inline-asm", "-Wno-unused-function", "-S", "--cuda-device-only",
                            "-I", os.path.dirname(os.path.abspath(path)), "-o", out, path], capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        path = out
    text = open(path).read()
    for name, body in kernels(text):
        if want not in name:
            continue
        print(name[:110])
        seen = set()
        for a, b in loops(body):
            n, gen, cyc = price(body[a:b + 1])
            m = n.get("mfma", 0)
            if m < 16 or (m, round(cyc)) in seen:
                continue
            seen.add((m, round(cyc)))
            others = ", ".join(f"{k} {v}" for k, v in sorted(n.items()) if k != "mfma")
            g = sum(gen.values())
            print(f"  loop of {b - a:5d} lines: {m:4d} MFMAs ({32 * m} cycles of the pipe) | {others} | compiler-generated outside the statements: {g}"
                  f" | issue sum {cyc:7.0f} cycles = {cyc / m:5.1f} per MFMA")


if __name__ == "__main__":
    main()
