#!/usr/bin/env python3
"""Audit of the compiled one-wave-per-SIMD forward (fa_fwd_w4_gfx950.hip): the instruction streams of fa_fwd_w4_asm.inc name
their registers literally, so hipcc must keep out of them.  Compiles the file to assembly and checks, per kernel:
  * no scratch (private_segment_fixed_size 0, vgpr_spill_count 0): a spill reload's vmcnt(0) would drain the LDS-DMA queue, and
    a spill INTO an accumulator register would corrupt O / Q / K silently;
  * no v_accvgpr_* and no scratch_* instruction outside the ;;#ASMSTART / ;;#ASMEND brackets;
  * hipcc's own VGPRs stay below the generator's budget NV (amdgpu_num_vgpr), and its own SGPRs below NS: the streams keep the K / V
    descriptors, cursors and the LDS base in literal s[NS:NS+13] across statements (amdgpu_num_sgpr);
  * the plain tile step (the statements that carry an LDS-DMA piece) is free of v_readlane / v_writelane (SGPR spills) and of
    compiler-made s_waitcnt vmcnt;
  * the hazards the inline-asm statements must keep by themselves (lint_blocks: v_exp_f32 -> next reader, MFMA result -> reader
    right behind it, M0 write -> LDS-DMA request, packed P -> MFMA).
    python tools/audit_w4.py [--keep DIR]      exit code 0 = clean
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_fwd_w4_gfx950.hip")


def compile_to_asm(outdir):
    out = os.path.join(outdir, "w4.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-inline-asm", "-S", "--cuda-device-only", "-o", out, SRC]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def kernels(text):
    """name -> list of lines of the kernel's code"""
    res = {}
    cur = None
    for l in text.split("\n"):
        m = re.match(r"^(_ZN8aule_hip\S*fa_fwd_w4_kernel\S*):", l)
        if m:
            cur = m.group(1)
            res[cur] = []
            continue
        if cur is not None:
            res[cur].append(l)
            if "s_endpgm" in l:
                cur = None
    return res


def _regs(tok):
    """register numbers of one operand: v5 -> {('v', 5)}, a[0:15] -> {('a', 0) .. ('a', 15)}"""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def lint_blocks(blocks):
    """Hazards inline asm has to keep by itself (hipcc's hazard recogniser does not look inside), checked on the instruction
    order of the COMPILED statements:
      * the result of a v_exp_f32 (transcendental unit) is not read by the very next instruction;
      * an MFMA's result is not read by a VALU / LDS / memory instruction within the next 2 instructions of the same statement
        (the streams keep every such reader at least 8 MFMAs away; this catches a placement bug, not a cycle count);
      * one instruction sits between an M0 write and the buffer_load ... lds that uses it;
      * a v_cvt_pk result (packed P) is not consumed by an MFMA in the next instruction."""
    out = []
    for bi, blk in enumerate(blocks):
        ins = [t for t in blk if t and not t.startswith(";")]
        parsed = []
        for t in ins:
            parts = t.replace(",", " ").split()
            op, ops = parts[0], parts[1:]
            dst = _regs(ops[0]) if ops and (op.startswith("v_") or op.startswith("ds_read") or (op.startswith("buffer_load") and "lds" not in t)) else set()
            src = set()
            for o in ops[1:] if dst else ops:
                src |= _regs(o)
            if op.startswith("v_mfma"):      # D, A, B, C: all but the first are sources
                dst = _regs(ops[0]); src = set().union(*[_regs(o) for o in ops[1:4]])
            parsed.append((op, dst, src, t))
        for i, (op, dst, src, t) in enumerate(parsed):
            nxt = parsed[i + 1:i + 3]
            if op == "v_exp_f32" and nxt and (dst & nxt[0][2]):
                out.append(f"asm statement {bi}: `{nxt[0][3]}` reads the result of `{t}` in the next instruction")
            if op.startswith("v_mfma"):
                for (op2, d2, s2, t2) in nxt:
                    if not op2.startswith("v_mfma") and (dst & s2):
                        out.append(f"asm statement {bi}: `{t2}` reads the result of `{t}` right behind it")
            if op == "s_add_u32" and "m0" in t.split(",")[0] and nxt and "lds" in nxt[0][3] and nxt[0][0].startswith("buffer_load"):
                out.append(f"asm statement {bi}: `{nxt[0][3]}` directly behind the M0 write")
            if op.startswith("v_cvt_pk") and nxt and nxt[0][0].startswith("v_mfma") and (dst & nxt[0][2]):
                out.append(f"asm statement {bi}: `{nxt[0][3]}` consumes `{t}` in the next instruction")
    return out[:20]


def lint_lds_waits(block, entry_pending=0):
    """In-order check of the LDS waits of ONE compiled asm statement whose reads are all consumed inside it (the one-wave-per-SIMD dQ
    kernel's iterations, tools/gen_dq4.py): every instruction that reads a register loaded by a ds_read of the statement must sit
    behind an s_waitcnt lgkmcnt(n) that guarantees the read has returned (LDS returns in order: after lgkmcnt(n) all but the newest n
    reads are back).  entry_pending: reads issued by the previous statement that may still be out at entry (counted, registers
    unknown: they can only make a wait stricter than needed, never looser).  Returns a list of problems."""
    out = []
    issued = entry_pending          # LDS reads issued so far (including the unknown ones at entry)
    done = 0                        # reads known complete
    loaded = {}                     # register -> index of the read that loads it
    for t in block:
        if not t or t.startswith(";"):
            continue
        parts = t.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                done = max(done, issued - int(m.group(1)))
            continue
        if op.startswith("ds_read"):
            for r in _regs(ops[0]):
                loaded[r] = issued
            issued += 1
            continue
        if op.startswith("v_"):
            srcs = ops[1:4] if op.startswith("v_mfma") else (ops if op.startswith("v_cmp") else ops[1:])
            for o in srcs:
                for r in sorted(_regs(o)):          # sorted: the register named in a message must not depend on the hash seed
                    if r in loaded and loaded[r] >= done:
                        out.append(f"`{t}` reads {r[0]}{r[1]} before its ds_read (number {loaded[r]} of {issued}) is known to be back")
                        break
            if not op.startswith("v_cmp"):
                for r in sorted(_regs(ops[0])):
                    if r in loaded and loaded[r] >= done:
                        out.append(f"`{t}` writes {r[0]}{r[1]} while a ds_read into it may still be out")
                    loaded.pop(r, None)
    return out[:10]


def generator_ns():
    """NS of the committed streams (fa_fwd_w4_asm.inc: `static constexpr int ... NS = 88`): the first literal scalar register"""
    inc = open(os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_fwd_w4_asm.inc")).read()
    m = re.search(r"\bNS = (\d+)", inc)
    return int(m.group(1)) if m else 88


def audit(path, verbose=True):
    NS = generator_ns()
    text = open(path).read()
    problems = []
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\S+:.*\n)+)", text):
        body = m.group(2)
        def field(k):
            mm = re.search(r"\." + k + r":\s+(\d+)", body)
            return int(mm.group(1)) if mm else None
        meta[m.group(1)] = {k: field(k) for k in ("private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "agpr_count")}
    ks = kernels(text)
    if not ks:
        problems.append("no fa_fwd_w4 kernel found in the assembly")
    for name, lines in ks.items():
        md = meta.get(name, {})
        d = 128 if "d128" in name else 64
        nv = 52 if d == 128 else 84
        if md.get("private_segment_fixed_size") != 0 or md.get("vgpr_spill_count") != 0:
            problems.append(f"{name}: scratch {md.get('private_segment_fixed_size')} bytes, {md.get('vgpr_spill_count')} VGPR spills")
        # split into asm blocks and the compiler code between them
        blocks, between, cur, inasm = [], [[]], [], False
        for l in lines:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                inasm, cur = True, []
                continue
            if t.startswith(";;#ASMEND"):
                inasm = False
                blocks.append(cur)
                between.append([])
                continue
            if inasm:
                cur.append(t)
            elif t and not t.startswith(";") and not t.startswith("."):
                between[-1].append(t)
        for seg in between:
            for t in seg:
                if "v_accvgpr" in t or t.startswith("scratch_"):
                    problems.append(f"{name}: compiler-made `{t}`")
                for r in re.findall(r"\bv(\d+)\b", t) + [x for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t) for x in (a, b)]:
                    if int(r) >= nv:
                        problems.append(f"{name}: compiler instruction touches v{r} >= NV {nv}: `{t}`")
                # the streams keep K / V descriptors, cursors and the LDS base in literal s[NS:NS+13] ACROSS statements: hipcc's
                # own code must stay below NS (amdgpu_num_sgpr keeps it there today; a toolchain bump that does not would corrupt the
                # descriptors silently).  Only the asm statements may name s >= NS.
                for r in re.findall(r"\bs(\d+)\b", t) + [x for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", t) for x in (a, b)]:
                    if int(r) >= NS:
                        problems.append(f"{name}: compiler instruction touches s{r} >= NS {NS}: `{t}`")
        problems += [f"{name}: {p}" for p in lint_blocks(blocks)]
        for blk in blocks:
            for t in blk:
                for r in re.findall(r"\bs(\d+)\b", t) + [x for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", t) for x in (a, b)]:
                    if int(r) >= NS + 14 and int(r) < 102:
                        problems.append(f"{name}: a statement names s{r}, outside the streams' s[{NS}:{NS + 13}]: `{t}`")
        # plain tile steps: eight consecutive statements with MFMAs and embedded LDS-DMA pieces, the tile barrier inside the first
        has_mf = [any("v_mfma" in b for b in blk) for blk in blocks]
        has_dma = [any("offen lds" in b for b in blk) for blk in blocks]
        plain_outside, nsteps = [], 0
        k = 0
        while k + 8 <= len(blocks):
            # (the step's wait + barrier rides inside its first statement)
            # (D = 128: a piece in every statement; D = 64: in every other one)
            if all(has_mf[k:k + 8]) and has_dma[k] and sum(has_dma[k:k + 8]) >= 4 and any("s_barrier" in b for b in blocks[k]):
                nsteps += 1
                for i in range(k + 1, k + 8):
                    plain_outside += between[i]
                k += 8
            else:
                k += 1
        if nsteps == 0:
            problems.append(f"{name}: no plain tile step recognised")
        bad = [t for t in plain_outside if t.startswith("v_readlane") or t.startswith("v_writelane") or (t.startswith("s_waitcnt") and "vmcnt" in t)]
        if bad:
            problems.append(f"{name}: plain tile steps contain {len(bad)} of v_readlane / v_writelane / s_waitcnt vmcnt, e.g. `{bad[0]}`")
        if verbose:
            nsalu = sum(1 for t in plain_outside if t.startswith("s_"))
            nvalu = sum(1 for t in plain_outside if t.startswith("v_"))
            print(f"{name[:90]}: {md}  {nsteps} plain step bodies, compiler code per body: {nsalu / max(nsteps, 1):.0f} SALU, {nvalu / max(nsteps, 1):.0f} VALU")
    return problems


if __name__ == "__main__":
    keep = None
    if len(sys.argv) > 2 and sys.argv[1] == "--keep":
        keep = sys.argv[2]
        os.makedirs(keep, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        path = compile_to_asm(keep or td)
        probs = audit(path)
    for pr in probs[:40]:
        print("PROBLEM:", pr)
    print("clean" if not probs else f"{len(probs)} problem(s)")
    sys.exit(0 if not probs else 1)
