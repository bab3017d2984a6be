#!/usr/bin/env python3
"""Writes aule-attention_amd/csrc/fa_bwd_dq4_asm.inc: the instruction streams of the one-wave-per-SIMD dQ kernel
(fa_bwd_dq4_gfx950.hip).  Run it after editing; the output is committed (the build does not need Python).

Workgroup = 4 waves, one per SIMD; a wave owns 64 query rows (two 32-row blocks rb = 0, 1) of a 256-row Q block, keeps their Q^T /
dO^T fragments and dQ^T in the accumulator file for the whole part, and walks the stream of 32-key KV blocks that the Q block sees.
Per KV block b (lane = query row, registers = keys -- the transposed forms, as in the forward):

    S^T_b  = K_b Q^T        dP^T_b = V_b dO^T          (2 x 16 MFMAs: A = row-major fragments of K_b / V_b, read just in time -- each
                                                        ds_read_b128 feeds both row blocks --, B = the resident fragments)
    P = exp2(c S - L')      dS = P (dP - delta)         (2 x 16 scores per lane; L', delta are per-lane constants of the part)
    dQ^T  += K_b^T dS^T                                 (2 x 8 MFMAs: A = transposed fragments of K_b, B = packed dS)

software-pipelined three deep: iteration j is ONE asm statement

    s_waitcnt vmcnt(2 NP); s_barrier                     (blocks <= j + 1 have landed for everybody)
    S / dP of block j      |  arithmetic of block j - 1  |  dQ of block j - 2  (its 16 transpose reads early in the statement)
    the LDS-DMA requests of block j + 4, the first two k-slices of block j + 1's fragments for the next iteration

with every part optional (QK, AR, DQ flags: stream start / tail, and a wave whose rows end below the workgroup's last blocks goes
idle early -- barrier + requests only).  LDS waits are counted (LDS returns in order): place() inserts s_waitcnt lgkmcnt(n) in
front of the first reader of every read.

Register map (D = 128; every register of the loop is named literally, hipcc keeps v0 .. v(NV-1): amdgpu_num_vgpr):

    accumulator file                                      arch VGPRs
    a[0:127]    dQ^T: row block rb, d block d at          v[0:NV)  hipcc          THR (2)  masked: key threshold per row block
                a[64 rb + 16 d ..]                         SC (4)   L'[rb], delta[rb]        T (8)   temporaries
    a[128:191]  Q^T fragments (rb, ks) at 128+32rb+4ks      VR, KR (8 + 8)  row-major V_b / K_b fragments, two k-slices deep
    a[192:255]  dO^T fragments                             DS[2][2][8]     packed dS by block parity, row block
                                                           DP[2][2][16], S[2][2][16]   by block parity, row block
                                                           KT (32)  transposed K fragments of block j - 2: (kk, d) at + 4 (4 kk + d)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_w4 import emit_asm, vregs, aregs, tup   # noqa: E402

OUT = os.environ.get("DQ4_OUT", os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_bwd_dq4_asm.inc"))
# Timing experiments only (tools/dq4_variants.sh; results are garbage): DQ4_X = comma list of
#   nobar  no barrier / vmcnt wait     novalu  no arithmetic     nolds  no LDS reads (stale fragments), no lgkmcnt waits
#   nodma  no LDS-DMA requests         nodq    no dQ MFMAs       nowait no lgkmcnt waits
XFLAGS = set(x for x in os.environ.get("DQ4_X", "").split(",") if x)


class Cfg:
    def __init__(self, D, dt):
        assert D in (64, 128)
        self.D, self.dt = D, dt
        self.RB, self.KS, self.DB = 2 * D, D // 16, D // 32
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
        # the dK/dV kernel's image of a 32-row block (tools/gen_bw4.py: D = 128 padded sub-tiles, D = 64 a chunk permutation per
        # 1 KB piece; one image serves ds_read_b128 and the transpose reads): K_b at 0, V_b at IMG of the block's ring slot
        if D == 128:
            self.PBASE = [1024 * rg + (0, 16, 128, 144)[rg & 3] + 256 * (rg >> 2) for rg in range(8)]
            self.IMG = 8704
            self.NP = 4                              # LDS-DMA pieces per wave and block (row groups 2 w, 2 w + 1 of K and of V)
        else:
            self.PBASE = [1040 * p for p in range(4)]
            self.IMG = 4352
            self.NP = 2                              # piece w (rows 8 w .. 8 w + 7) of K and of V
        self.SLOT = 2 * self.IMG
        self.DQ = 0                                  # accumulator file: dQ^T (2 row blocks x DB x 16), Q^T, dO^T fragments (2 x KS x 4 each)
        self.QF = self.DQ + 32 * self.DB
        self.DF = self.QF + 8 * self.KS
        self.KT = 256 - 8 * self.DB                  # arch VGPRs, top down
        self.S = self.KT - 64
        self.DP = self.S - 64
        self.DS = self.DP - 32
        self.KR = self.DS - 8
        self.VR = self.KR - 8
        self.T = self.VR - 8
        self.SC = self.T - 4
        self.THR = self.SC - 2
        self.NV = self.THR

    def s(self, par, rb):
        return self.S + 32 * par + 16 * rb

    def dp(self, par, rb):
        return self.DP + 32 * par + 16 * rb

    def ds(self, par, rb):
        return self.DS + 16 * par + 8 * rb


def crow(r):
    """key (minus 4 hi) inside the 32-key block of accumulator register r"""
    return (r & 3) + 8 * (r >> 2)


def arith_ops(c, par, rb, q, masked):
    """dS of scores 4 q .. 4 q + 3 of row block rb, block parity par -> two packed registers."""
    S, DPr, t = c.s(par, rb), c.dp(par, rb), [c.T + i for i in range(8)]
    lp, dl = c.SC + rb, c.SC + 2 + rb
    r0 = 4 * q
    ops = []
    for i in range(4):
        ops.append(f"v_fma_f32 v{t[i]}, v{S + r0 + i}, %[c], -v{lp}")
    for i in range(4):
        ops.append(f"v_sub_f32 v{t[4 + i]}, v{DPr + r0 + i}, v{dl}")
    for i in range(4):
        ops.append(f"v_exp_f32 v{t[i]}, v{t[i]}")
    if masked == 3:
        # sliding window (round 5): THR[rb] holds (first visible key - 4 hi) - k0 here, %[wd{rb}] the number of visible keys of the lane's row:
        # key crow + 4 hi of the block is visible iff (unsigned)(crow - THR) < wd.  The temporary is a register of this block's dS tuple
        # (written only by the pack at the end of the group)
        tm = c.ds(par, rb) + r0 // 2
        for i in range(4):
            ops.append(f"v_sub_u32 v{tm}, {crow(r0 + i)}, v{c.THR + rb}")
            ops.append(f"v_cmp_gt_u32 vcc, %[wd{rb}], v{tm}")
            ops.append(f"v_cndmask_b32 v{t[i]}, 0, v{t[i]}, vcc")
    elif masked:
        for i in range(4):
            ops.append(f"v_cmp_le_i32 vcc, {crow(r0 + i)}, v{c.THR + rb}")
            ops.append(f"v_cndmask_b32 v{t[i]}, 0, v{t[i]}, vcc")
    for i in range(4):
        ops.append(f"v_mul_f32 v{t[4 + i]}, v{t[i]}, v{t[4 + i]}")
    for j in range(2):
        ops.append(f"{c.cvt} v{c.ds(par, rb) + r0 // 2 + j}, v{t[4 + 2 * j]}, v{t[5 + 2 * j]}")
    return ops


def tr_reads(c, kk, d):
    """the two transpose reads of (16-key step kk, d block d) of block j - 2's K image -> KT + 4 (4 kk + d) .. + 3"""
    b = c.KT + 4 * (c.DB * kk + d)
    if c.D == 128:
        off, o2 = (c.PBASE[2 * (2 * kk + e)] + 256 * d for e in (0, 1))
    else:
        off, o2 = (c.PBASE[2 * kk + e] + 512 * d for e in (0, 1))
    return [f"ds_read_b64_tr_b16 v[{b}:{b + 1}], %[trb] offset:{off}", f"ds_read_b64_tr_b16 v[{b + 2}:{b + 3}], %[trb] offset:{o2}"]


def rm_reads(c, ks, base):
    """row-major fragments of k-slice ks (d = 16 ks + 8 hi ..) of a block: K -> KR, V -> VR (buffer ks & 1)"""
    if c.D == 128:
        off = 128 * ks
    else:       # k-slice ks = (d-slice ks >> 1, b = ks & 1): the lane base of that b ("%[ra]" -> "%[rab]"), + 512 d
        off = 512 * (ks >> 1)
        if ks & 1:
            base = base[:-1] + "b]"
    return [f"ds_read_b128 v[{c.KR + 4 * (ks & 1)}:{c.KR + 4 * (ks & 1) + 3}], {base} offset:{off}",
            f"ds_read_b128 v[{c.VR + 4 * (ks & 1)}:{c.VR + 4 * (ks & 1) + 3}], {base} offset:{c.IMG + off}"]


def regs_of(tok):
    import re
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


MERGE_KV = os.environ.get("DQ4_MERGE_KV", "1") != "0"      # one wait per k-slice for its K AND V fragment (A/B: 0 = one each)


def with_waits(lines, pending, pair=None):
    """insert s_waitcnt lgkmcnt(n) in front of the first reader of every LDS read.  pending: registers loaded by reads that were
    issued before the statement, oldest first (list of register sets); LDS returns in order.  pair: (lo, hi, delta) -- a reader of a
    register in [lo, hi) also waits for register + delta (the S MFMAs of a k-slice wait for its V fragment too: the dP MFMAs two
    slots later then need no s_waitcnt of their own -- one instruction per k-slice less for a read that was issued two MFMAs
    after the K fragment's and six MFMAs ago)."""
    out = []
    issued = []                       # register sets of the LDS reads in issue order (pending first)
    done = 0                          # reads known complete: issued[:done]
    issued += pending
    for ln in lines:
        parts = ln.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if op.startswith("ds_read"):
            issued.append(regs_of(ops[0]))
            out.append(ln)
            continue
        if op.startswith("v_"):
            src = set()
            for o in (ops[1:] if not op.startswith("v_cmp") else ops):
                src |= regs_of(o)
            if op.startswith("v_mfma"):
                src = set().union(*[regs_of(o) for o in ops[1:4]])
                if pair is not None:
                    src |= {(f, r + pair[2]) for (f, r) in src if f == "v" and pair[0] <= r < pair[1]}
            need = -1
            for i in range(done, len(issued)):
                if issued[i] & src:
                    need = i
            if need >= 0:
                n = min(len(issued) - 1 - need, 15)     # (the counter has four bits: at most 15 may stay out)
                out.append(f"s_waitcnt lgkmcnt({n})")
                done = len(issued) - n
        out.append(ln)
    return out, issued[done:]


def gen_iter(c, par, qk, nxt, ar, dq, pre):
    """iteration j with block parity par = j & 1.  qk: S / dP of block j (pre: its first two k-slices were requested by the previous
    statement); nxt: request the first two k-slices of block j + 1; ar: 0 / 1 / 2 (masked) arithmetic of block j - 1; dq: dQ of
    block j - 2 (+ its transpose reads)."""
    mf = []          # (mfma line, pinned fillers behind it)
    if qk:
        for ks in range(c.KS):
            grp = []
            for rb in (0, 1):
                s = tup(c.s(par, rb), 16)
                grp.append(f"{c.mfma} {s}, v[{c.KR + 4 * (ks & 1)}:{c.KR + 4 * (ks & 1) + 3}], a[{c.QF + 4 * c.KS * rb + 4 * ks}:{c.QF + 4 * c.KS * rb + 4 * ks + 3}], {'0' if ks == 0 else s}")
            for rb in (0, 1):
                d = tup(c.dp(par, rb), 16)
                grp.append(f"{c.mfma} {d}, v[{c.VR + 4 * (ks & 1)}:{c.VR + 4 * (ks & 1) + 3}], a[{c.DF + 4 * c.KS * rb + 4 * ks}:{c.DF + 4 * c.KS * rb + 4 * ks + 3}], {'0' if ks == 0 else d}")
            for i, m in enumerate(grp):
                # behind the last MFMA that reads a buffer of k-slice ks: the read of k-slice ks + 2 into it (K behind the two S
                # MFMAs, V behind the two dP MFMAs: six MFMAs ahead of its first reader)
                pin = []
                if ks + 2 < c.KS and i in (1, 3):
                    pin = [rm_reads(c, ks + 2, "%[ra]")[0 if i == 1 else 1]]
                mf.append((m, pin))
    ndq0 = len(mf)
    if dq:
        for kk in range(2):
            for d in range(c.DB):
                for rb in (0, 1):
                    acc = f"a[{c.DQ + 16 * c.DB * rb + 16 * d}:{c.DQ + 16 * c.DB * rb + 16 * d + 15}]"
                    kt = c.KT + 4 * (c.DB * kk + d)
                    b = c.ds(par, rb) + 4 * kk       # block j - 2 has parity par
                    mf.append((f"{c.mfma} {acc}, v[{kt}:{kt + 3}], v[{b}:{b + 3}], {acc}", []))
    # floating fillers, in program order: transpose reads first (their MFMAs come last), then the arithmetic
    fl = []
    trs = []
    if dq:
        for kk in range(2):
            for d in range(c.DB):
                trs += tr_reads(c, kk, d)
    if ar:
        for rb in (0, 1):
            for q in range(4):
                fl += arith_ops(c, par ^ 1, rb, q, ar if ar >= 2 else 0)
    head = [f"s_waitcnt vmcnt({2 * c.NP})", "s_barrier"]      # all but the two newest blocks' pieces of this wave: blocks <= j + 1 have landed
    if ar >= 2:
        head += [f"v_subrev_u32 v{c.THR}, %[k0], %[lim0]", f"v_subrev_u32 v{c.THR + 1}, %[k0], %[lim1]"]   # thr = lim - k0 (window: lim = the first visible key)
    if qk and not pre:
        head += rm_reads(c, 0, "%[ra]") + rm_reads(c, 1, "%[ra]")
    if dq and not qk:
        head += trs            # (stream tail: nothing to hide them behind)
    # tail: the LDS-DMA requests of block j + 4 and the next block's first fragments
    dma = []
    for img in (0, 1):
        for half in range(c.NP // 2):
            dma.append((f"s_add_u32 m0, %[dlds], {img * c.IMG + half * 1040}",
                        f"buffer_load_dwordx4 %[vost{half}], {'%[ksrd]' if img == 0 else '%[vsrd]'}, %[dso] offen lds"))
    nx = (rm_reads(c, 0, "%[ra2]") + rm_reads(c, 1, "%[ra2]")) if nxt else []
    lines = list(head)
    n = len(mf)
    if n == 0:
        lines += fl
        for a, b in dma:
            lines += [a, "s_nop 0", b]
        lines += nx
    else:
        # the DMA pieces ride on the last four MFMAs (M0 write in front of the MFMA, request behind it); floating fillers evenly
        # over the gaps; the next block's fragments behind the last MFMA that reads KR / VR (or at the very end)
        k = 0
        for g, (m, pin) in enumerate(mf):
            di = g - (n - len(dma))
            if di >= 0:
                lines.append(dma[di][0])
            lines.append(m)
            if di >= 0:
                lines.append(dma[di][1])
            lines += pin
            if dq and qk and g < len(trs) // 2:
                lines += trs[2 * g:2 * g + 2]      # the transpose reads ride on the first eight gaps, 24 MFMAs ahead of their readers
            if nx and g == ndq0 - 1:
                lines += nx
            take = len(fl) * (g + 1) // n - len(fl) * g // n
            lines += fl[k:k + take]
            k += take
        if not dq:
            lines += ["s_nop 7", "s_nop 7"]          # the S / dP MFMAs are the statement's last: their readers follow closely
    pending = []
    if qk and pre:
        for ks in (0, 1):
            pending += [regs_of(f"v[{c.KR + 4 * ks}:{c.KR + 4 * ks + 3}]"), regs_of(f"v[{c.VR + 4 * ks}:{c.VR + 4 * ks + 3}]")]
    lines, left = with_waits(lines, pending, (c.KR, c.KR + 8, c.VR - c.KR) if MERGE_KV else None)
    if XFLAGS:
        def keep(ln):
            op = ln.split()[0]
            if "nobar" in XFLAGS and (op == "s_barrier" or ln.startswith("s_waitcnt vmcnt")):
                return False
            if "novalu" in XFLAGS and op.startswith("v_") and not op.startswith("v_mfma"):
                return False
            if ("nolds" in XFLAGS or "nowait" in XFLAGS) and ln.startswith("s_waitcnt lgkmcnt"):
                return False
            if "nolds" in XFLAGS and op.startswith("ds_read"):
                return False
            if "nodma" in XFLAGS and (op == "s_add_u32" or (op.startswith("buffer_load") and ln.endswith("lds"))):
                return False
            if "nodq" in XFLAGS and op.startswith("v_mfma") and ln.split()[1].startswith("a["):
                return False
            return True
        lines = [ln for ln in lines if keep(ln)]
    assert len(left) == (4 if nxt else 0), (len(left), nxt)
    assert not nxt or qk
    clob = ["memory", "m0", "scc"]
    if qk:
        clob += vregs(c.S + 32 * par, 32) + vregs(c.DP + 32 * par, 32)
    if qk or nxt:
        clob += vregs(c.VR, 16)
    if ar:
        clob += vregs(c.T, 8) + vregs(c.DS + 16 * (par ^ 1), 16)
        if ar >= 2:
            clob += ["vcc"] + vregs(c.THR, 2)
    if dq:
        clob += vregs(c.KT, 8 * c.DB) + aregs(c.DQ, 32 * c.DB)
    ins = ['[dlds] "s"(dlds)', '[ksrd] "s"(ksrd)', '[vsrd] "s"(vsrd)', '[dso] "s"(dso)', '[vost0] "v"(vost0)', '[vost1] "v"(vost1)']
    if qk:
        ins.append('[ra] "v"(ra)')
        if c.D == 64:
            ins.append('[rab] "v"(rab)')
    if nxt:
        ins.append('[ra2] "v"(ra2)')
        if c.D == 64:
            ins.append('[ra2b] "v"(ra2b)')
    if ar:
        ins.append('[c] "s"(c)')
        if ar >= 2:
            ins += ['[lim0] "v"(lim0)', '[lim1] "v"(lim1)', '[k0] "s"(k0)']
        if ar == 3:
            ins += ['[wd0] "v"(wd0)', '[wd1] "v"(wd1)']
    if dq:
        ins.append('[trb] "v"(trb)')
    return emit_asm(lines, [], ins, clob)


def gen_struct(c):
    name = f"Dq4Asm<{'Bf16Traits' if c.dt == 'bf16' else 'F16Traits'}, {c.D}>"
    s = f"template <> struct {name} {{\n"
    s += (f"    static constexpr int NV = {c.NV}, NP = {c.NP}, SLOT = {c.SLOT}, IMG = {c.IMG}, PB1 = {c.PBASE[1]}, PB2 = {c.PBASE[2]}, PB4 = {c.PBASE[4] if len(c.PBASE) > 4 else 0};\n"
          f"    static constexpr int SC = {c.SC}, QF = {c.QF}, DF = {c.DF};   // NV: hipcc's VGPR budget (amdgpu_num_vgpr)\n")
    s += ("    // iteration j (PAR = j & 1): S / dP of block j (QK; PRE: its first fragments were requested by the previous statement),\n"
          "    // arithmetic of block j - 1 (AR: 1 plain, 2 masked, 3 window-masked: lim = first visible key, wd = visible keys), dQ of block j - 2 (DQ), first fragments of block j + 1 (NXT)\n"
          "    template <int PAR, int QK, int NXT, int AR, int DQ, int PRE>\n"
          "    static __device__ __forceinline__ void iter(float c, unsigned ra, unsigned rab, unsigned ra2, unsigned ra2b, unsigned trb, int lim0, int lim1, int k0, unsigned dlds,\n"
          "                                                __amdgpu_buffer_rsrc_t ksrd, __amdgpu_buffer_rsrc_t vsrd, unsigned dso, unsigned vost0, unsigned vost1, int wd0 = 0, int wd1 = 0) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)c; (void)ra; (void)rab; (void)ra2; (void)ra2b; (void)trb; (void)lim0; (void)lim1; (void)k0; (void)wd0; (void)wd1;\n"
          "        dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n        dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n"
          "        k0 = __builtin_amdgcn_readfirstlane(k0);\n")
    first = True
    # j = 0: S / dP only; j = 1: + arithmetic; j >= 2: + dQ; j = n_w: no S / dP any more; n_w + 1: dQ only; then idle
    variants = [(0, 1, nxt, 0, 0, 0) for nxt in (0, 1)]
    variants += [(par, 1, nxt, ar, dq, 1) for par in (0, 1) for nxt in (0, 1) for ar in (1, 2, 3) for dq in (0, 1)]
    variants += [(par, 0, 0, ar, dq, 0) for par in (0, 1) for ar in (0, 1, 2, 3) for dq in (0, 1)]
    for (par, qk, nxt, ar, dq, pre) in variants:
        s += (f"        {'if' if first else 'else if'} constexpr (PAR == {par} && QK == {qk} && NXT == {nxt} && AR == {ar} && DQ == {dq} && PRE == {pre}) {{\n")
        s += gen_iter(c, par, qk, nxt, ar, dq, pre) + "        }\n"
        first = False
    s += "        else static_assert(PAR < 0, \"fa_bwd_dq4_asm.inc: variant not generated\");\n#endif\n    }\n"
    # ---- Q^T / dO^T fragments of the wave's 64 rows straight into the accumulator file: lane (row, hi) holds d = 16 ks + 8 hi .. + 7
    # of its row (rows >= Sq read as 0).  vo0 / vo1: byte offset of the lane's row of row block 0 / 1 plus 16 hi
    lines = ["s_nop 4"]
    for rb in (0, 1):
        for ks in range(c.KS):
            lines.append(f"buffer_load_dwordx4 a[{c.QF + 4 * c.KS * rb + 4 * ks}:{c.QF + 4 * c.KS * rb + 4 * ks + 3}], %[vo{rb}], %[qsrd], 0 offen offset:{32 * ks}")
            lines.append(f"buffer_load_dwordx4 a[{c.DF + 4 * c.KS * rb + 4 * ks}:{c.DF + 4 * c.KS * rb + 4 * ks + 3}], %[vo{rb}], %[gsrd], 0 offen offset:{32 * ks}")
            # (the same fragments of O, for delta = rowsum(O * dO), into the dQ accumulators: zeroed afterwards)
            lines.append(f"buffer_load_dwordx4 a[{4 * c.KS * rb + 4 * ks}:{4 * c.KS * rb + 4 * ks + 3}], %[vo{rb}], %[osrd], 0 offen offset:{32 * ks}")
    lines.append("s_waitcnt vmcnt(0)")
    s += ("    static __device__ __forceinline__ void load_frags(__amdgpu_buffer_rsrc_t qsrd, __amdgpu_buffer_rsrc_t gsrd, __amdgpu_buffer_rsrc_t osrd, unsigned vo0, unsigned vo1) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n")
    s += emit_asm(lines, [], ['[qsrd] "s"(qsrd)', '[gsrd] "s"(gsrd)', '[osrd] "s"(osrd)', '[vo0] "v"(vo0)', '[vo1] "v"(vo1)'], ["memory"] + aregs(c.QF, 16 * c.KS) + aregs(0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- accumulators to zero
    lines = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(32 * c.DB)]
    s += "    static __device__ __forceinline__ void zero_acc() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], aregs(0, 32 * c.DB), indent="        ")
    s += "#endif\n    }\n"
    # ---- the LDS-DMA pieces of one block as a statement of its own (stream start)
    lines = ["s_nop 4"]
    for img in (0, 1):
        for half in range(c.NP // 2):
            lines += [f"s_add_u32 m0, %[dlds], {img * c.IMG + half * 1040}", "s_nop 0",
                      f"buffer_load_dwordx4 %[vost{half}], {'%[ksrd]' if img == 0 else '%[vsrd]'}, %[dso] offen lds"]
    s += ("    static __device__ __forceinline__ void dma_block(unsigned dlds, __amdgpu_buffer_rsrc_t ksrd, __amdgpu_buffer_rsrc_t vsrd, unsigned dso,\n"
          "                                                     unsigned vost0, unsigned vost1) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n        dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n")
    s += emit_asm(lines, [], ['[dlds] "s"(dlds)', '[ksrd] "s"(ksrd)', '[vsrd] "s"(vsrd)', '[dso] "s"(dso)',
                              '[vost0] "v"(vost0)', '[vost1] "v"(vost1)'], ["memory", "m0", "scc"], indent="        ")
    s += "#endif\n    }\n"
    # ---- the per-lane constants of a row block: L' = LSE log2(e), delta
    for rb in range(2):
        s += (f"    static __device__ __forceinline__ void set_scal{rb}(float lp, float delta) {{\n#if defined(__HIP_DEVICE_COMPILE__)\n"
              f"        asm volatile(\"v_mov_b32 v{c.SC + rb}, %0\\n\\tv_mov_b32 v{c.SC + 2 + rb}, %1\" :: \"v\"(lp), \"v\"(delta) : \"v{c.SC + rb}\", \"v{c.SC + 2 + rb}\");\n"
              "#endif\n    }\n")
    s += "};\n\n"
    return s


def main():
    hdr = ("// fa_bwd_dq4_asm.inc -- GENERATED by tools/gen_dq4.py (do not edit; edit the generator and re-run it).\n"
           "// Instruction streams of the one-wave-per-SIMD dQ kernel: register map, pipeline and hazards in the generator's docstring.\n"
           "// Included by fa_bwd_dq4_gfx950.hip inside namespace aule_hip::{anonymous}.\n\n"
           "template <class T, int D> struct Dq4Asm;\n\n")
    body = ""
    for D in (128, 64):
        for dt in ("bf16", "fp16"):
            body += gen_struct(Cfg(D, dt))
    with open(OUT, "w") as fh:
        fh.write(hdr + body)
    for D in (128, 64):
        c = Cfg(D, "bf16")
        print(f"D={D}: NV={c.NV} THR={c.THR} SC={c.SC} T={c.T} VR={c.VR} KR={c.KR} DS={c.DS} DP={c.DP} S={c.S} KT={c.KT}; acc DQ={c.DQ} QF={c.QF} DF={c.DF}")
    print(f"wrote {OUT}: {len((hdr + body).splitlines())} lines; NV={c.NV} THR={c.THR} SC={c.SC} T={c.T} VR={c.VR} KR={c.KR} DS={c.DS} DP={c.DP} S={c.S} KT={c.KT}")


if __name__ == "__main__":
    main()
