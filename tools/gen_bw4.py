#!/usr/bin/env python3
"""Writes aule-attention_amd/csrc/fa_bwd_dkv4_asm.inc: the instruction streams of the one-wave-per-SIMD dK/dV kernel
(fa_bwd_dkv4_gfx950.hip).  Run it after editing; the output is committed (the build does not need Python).

Workgroup = 4 waves, one per SIMD; a wave owns 32 key rows of a 128-key KV block and the whole 512-register file, and walks a
stream of 32-row query blocks (all query heads of the GQA group, every block that sees the KV block).  Per block b:

    S_b  = Q_b K^T        dP_b = dO_b V^T                  (16 MFMAs: A = row-major fragments of Q_b / dO_b, B = K / V fragments)
    P_b  = exp2(c S_b - L'_b)      dS_b = P_b (dP_b - delta_b)              (16 scores per lane; lane = key, registers = rows;
                                                                             - delta_b is the C operand of dP_b's first MFMA)
    dV^T += dO_b^T P_b    dK^T += Q_b^T dS_b               (16 MFMAs: A = transposed fragments of dO_b / Q_b, B = packed P / dS)

software-pipelined over the stream: iteration i is

    phase 1   S_{i+1}, dP_{i+1}   |  the arithmetic of block i  |  32 transpose reads of block i
    ---- s_waitcnt vmcnt(NP) lgkmcnt(0); s_barrier ----       (block i + 2 has landed; everybody is done with block i's images)
    phase 2   dV, dK of block i   |  16 row-major reads of block i + 2 (into accumulator registers)  |  L', delta of block i + 2
                                     requested, L' of block i + 1 scaled  |  the LDS-DMA requests of block i + 4 (its slot = block i's)

Register map (D = 128; every register of the loop is named literally, hipcc keeps v0 .. v(NV-1): amdgpu_num_vgpr):

    accumulator file                                arch VGPRs
    a[0:63]     dV^T, block d at a[16 d ..]         v[0:NV)     hipcc
    a[64:127]   dK^T                                T  (8)      temporaries
    a[128:159]  K fragments (ks at a[128 + 4 ks])   DL[2][16]   delta of the block, by block parity
    a[160:191]  V fragments                         LD[2][16]   L' = LSE log2(e), by block parity
    a[192:223]  Q_b row-major fragments (ks)        DS (8), P (8)   packed dS, P: B operands of phase 2
    a[224:255]  dO_b row-major fragments            DP[2][16], S[2][16]   by block parity
                                                    Y (32), X (32)  transposed fragments of Q_b / dO_b: step st at + 4 st

Hazards kept by the strings themselves: v_exp_f32 result -> reader at least one instruction later; an M0 write and its request
one instruction apart; everything an MFMA reads from a VALU / LDS result sits behind the phase boundary's waits.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_w4 import emit_asm, vregs, aregs, tup   # noqa: E402

OUT = os.environ.get("BW4_OUT", os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_bwd_dkv4_asm.inc"))
# SPILL instances (the 5-matmul backward): where the two dS stores of an iteration sit -- "start": at the top of phase 2, in front of the
# iteration's other VMEM requests (the phase boundary's vmcnt(NP) then waits for them too); "end": behind the LDS-DMA pieces, the boundary
# waits with vmcnt(NP + 2) (the stores get two iterations to retire).  BW4_ST_NT=1: non-temporal stores.
# Timing / energy experiments only (tools/bw4_energy_variants.sh; results are garbage): BW4_X = comma list of
#   novalu  no P / dS arithmetic      nolds  no LDS reads (stale fragments)      nodma  no LDS-DMA requests      noscal  no L' / delta loads
XFLAGS = set(x for x in os.environ.get("BW4_X", "").split(",") if x)


def xfilter(lines):
    if not XFLAGS:
        return lines
    out = []
    for ln in lines:
        op = ln.split()[0]
        if "novalu" in XFLAGS and op.startswith("v_") and not op.startswith("v_mfma"):
            continue
        if "nolds" in XFLAGS and op.startswith("ds_read"):
            continue
        if "nodma" in XFLAGS and (ln.startswith("s_add_u32 m0") or (op.startswith("buffer_load") and ln.rstrip().endswith("lds"))):
            continue
        if "noscal" in XFLAGS and op.startswith("buffer_load") and not ln.rstrip().endswith("lds"):
            continue
        out.append(ln)
    return out or ["s_nop 0"]


TR64_EARLY = os.environ.get("BW4_TR64", "") == "early"
ST_LATE = 2 if os.environ.get("BW4_ST", "start") == "end" else 0
ST_NT = os.environ.get("BW4_ST_NT", "0") == "1"


class Cfg:
    def __init__(self, D, dt):
        assert D in (64, 128)
        self.D, self.dt = D, dt
        self.RB = 2 * D
        self.KS, self.DB = D // 16, D // 32
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
        if D == 128:
            # ONE image per tensor and 32-row block (round 3b; before: a swizzled row-major image for the ds_read_b128 fragments AND a
            # sub-tiled one for the transpose reads, twice the LDS-DMA pieces).  Piece rg = the four rows 4 rg .. 4 rg + 3 as eight
            # [4 rows][16 d] sub-tiles of 128 bytes (what a transpose read wants: a pass of 32 lanes covers two neighbouring
            # sub-tiles, 256 contiguous bytes); the pieces sit at PBASE[rg] = 1024 rg + {0, 16, 128, 144}[rg & 3] + 256 (rg >> 2):
            # a 16-lane pass of a ds_read_b128 (rows r .. r + 15 = four pieces, one 16-byte chunk each at (r & 3) 32 inside its
            # sub-tile) then covers all 64 banks once.  No linear piece stride does that; the pads cost 512 bytes per image.
            self.PBASE = [1024 * rg + (0, 16, 128, 144)[rg & 3] + 256 * (rg >> 2) for rg in range(8)]
            self.IMG = 8704                             # >= PBASE[7] + 1024, a multiple of 256
            self.NP = 4                                 # DMA pieces per wave and block: row groups 2 w, 2 w + 1 of Q and of dO
            self.RM_SPLIT = (2, 2, 6, 6)                # ds_read_b128 of block i + 2 per phase-2 statement (16: k-slices 0 .. 7 of Q, dO)
        else:
            # D = 64 (round 4): a row is 128 bytes, so a 1 KB LDS-DMA piece (lane-linear in LDS: lane l writes chunk l, 16 bytes)
            # holds TWO row groups -- piece p = rows 8 p .. 8 p + 7 -- and no pad can sit between them.  The global side of a
            # piece is per lane, so the ORDER of its 64 chunks is free: chunk (rgl = row group inside the piece, d-slice d = 32
            # columns, b = which 16 of them, rr = row inside the group, h = which 8 of the 16) sits at chunk64() below,
            #     32 d + 16 rgl + 8 (rgl ^ b) + 2 rr + (h ^ b),
            # and the pieces at PBASE[p] = 1040 p (one chunk of pad each).  Then (positions mod 16 = banks / 4):
            #   * a 16-lane pass of a ds_read_b128 -- rows r .. r + 15 = two pieces x two row groups x four rows at fixed (d, b, h) --
            #     takes 8 (rgl ^ b) + 2 rr + const from one piece and the same + 1 from the other: 16 different values;
            #   * a 32-lane pass of a transpose read -- one row group (rgl fixed), sub-tiles b = 0 and 1, rows rr, halves h --
            #     takes 8 (rgl ^ b) + 2 rr + (h ^ b): 16 different values.
            # b enters through an exclusive-or with lane bits, so the row-major reads use one lane base per b (ra / ra1) and
            # the immediate carries only image and d; the transpose reads know b from the lane (bit 4).
            self.PBASE = [1040 * p for p in range(4)]
            self.IMG = 4352                             # >= PBASE[3] + 1024, a multiple of 256
            self.NP = 2                                 # DMA pieces per wave and block: piece w of Q and of dO
            # 8 reads: k-slices 0 .. 3 of Q, dO.  BW4_RM64 (experiment): another split, e.g. 4,4,0,0 -- the reads early in the phase, so that the
            # lgkmcnt(0) behind statement 3 finds them done
            self.RM_SPLIT = tuple(int(x) for x in os.environ.get("BW4_RM64", "2,2,2,2").split(","))
            assert len(self.RM_SPLIT) == 4 and sum(self.RM_SPLIT) == 8
        self.SLOT = 2 * self.IMG                    # Q, dO
        # accumulator file
        self.DV = 0
        self.DK = self.DV + 16 * self.DB
        self.KF = self.DK + 16 * self.DB
        self.VF = self.KF + 4 * self.KS
        self.QA = self.VF + 4 * self.KS
        self.DA = self.QA + 4 * self.KS
        # arch VGPRs, top down
        self.NST = 2 * self.DB                      # steps of the dV / dK products (two 16-row k-steps x DB d-slices)
        self.X = 256 - 4 * self.NST
        self.Y = self.X - 4 * self.NST
        self.S = self.Y - 32
        self.DP = self.S - 32
        self.P = self.DP - 8
        self.DS = self.P - 8
        self.LD = self.DS - 32
        self.DL = self.LD - 32
        self.T = self.DL - 8
        self.NV = self.T

    def acc(self, base, i, n=16):
        return f"a[{base + i * n}:{base + i * n + n - 1}]"

    def frag(self, base, i, file="a"):
        return f"{file}[{base + 4 * i}:{base + 4 * i + 3}]"


def chunk64(rgl, d, b, rr, h):
    """D = 64: position (in 16-byte chunks) inside its 1 KB piece of rows 4 rgl + rr, columns 32 d + 16 b + 8 h .. + 7 (Cfg)"""
    return 32 * d + 16 * rgl + 8 * (rgl ^ b) + 2 * rr + (h ^ b)


def crow(r):
    """query row (minus 4 hi) inside the 32-row block of accumulator register r"""
    return (r & 3) + 8 * (r >> 2)


def arith_ops(c, par, q, masked):
    """P / dS of scores 4 q .. 4 q + 3 of block parity par.  Two pairs in flight; masked: a score outside [lo, lo + width) of its
    lane (row constant minus the lane's first valid row, unsigned compare) gets weight 0."""
    S, DPr, LD, DL = c.S + 16 * par, c.DP + 16 * par, c.LD + 16 * par, c.DL + 16 * par
    t = [c.T + i for i in range(8)]
    r0 = 4 * q
    ops = []
    for i in range(4):
        ops.append(f"v_fma_f32 v{t[i]}, v{S + r0 + i}, %[c], -v{LD + r0 + i}")
    for i in range(4):
        ops.append(f"v_exp_f32 v{t[i]}, v{t[i]}")
    if masked:
        tm = c.DS + r0 // 2            # (a register of this statement's dS pair: written only by the pack at the end)
        for i in range(4):
            ops.append(f"v_sub_u32 v{tm}, {crow(r0 + i)}, %[lo]")
            ops.append(f"v_cmp_gt_u32 vcc, %[wd], v{tm}")
            ops.append(f"v_cndmask_b32 v{t[i]}, 0, v{t[i]}, vcc")
    for i in range(4):
        ops.append(f"v_mul_f32 v{t[4 + i]}, v{t[i]}, v{DPr + r0 + i}")      # dS = P (dP - delta): the MFMA chain of dP started from - delta
    for j in range(2):
        ops.append(f"{c.cvt} v{c.P + r0 // 2 + j}, v{t[2 * j]}, v{t[2 * j + 1]}")
        ops.append(f"{c.cvt} v{c.DS + r0 // 2 + j}, v{t[4 + 2 * j]}, v{t[5 + 2 * j]}")
    return ops


def deal(mfmas, fillers):
    """MFMA g, then an even share of the fillers (program order kept)."""
    n = len(mfmas)
    out = []
    if n == 0:
        return list(fillers)
    k = 0
    for g in range(n):
        out.append(mfmas[g])
        take = len(fillers) * (g + 1) // n - len(fillers) * g // n
        out += fillers[k:k + take]
        k += take
    return out


def tr_reads(c, st):
    """the four transpose reads of step st (k-step st / DB = 16 query rows, d-slice st % DB) of the current block: dO -> X, Q -> Y"""
    kk, d = st // c.DB, st % c.DB
    if c.D == 128:
        # rows 16 kk + 8 e + 4 hi .. + 3 (row group 2 (2 kk + e) + hi: the lane's constant carries PBASE[hi]), d = 32 d ..
        off, o2 = (c.PBASE[2 * (2 * kk + e)] + 256 * d for e in (0, 1))
    else:
        # rows 16 kk + 8 e + 4 hi .. + 3 = piece 2 kk + e, row group hi of it (the lane's constant carries rgl = hi, b, rr, h)
        off, o2 = (c.PBASE[2 * kk + e] + 512 * d for e in (0, 1))
    return [f"ds_read_b64_tr_b16 v[{c.X + 4 * st}:{c.X + 4 * st + 1}], %[trb] offset:{c.IMG + off}",
            f"ds_read_b64_tr_b16 v[{c.X + 4 * st + 2}:{c.X + 4 * st + 3}], %[trb] offset:{c.IMG + o2}",
            f"ds_read_b64_tr_b16 v[{c.Y + 4 * st}:{c.Y + 4 * st + 1}], %[trb] offset:{off}",
            f"ds_read_b64_tr_b16 v[{c.Y + 4 * st + 2}:{c.Y + 4 * st + 3}], %[trb] offset:{o2}"]


def gen_p1(c, q, par, qk, ar, tr):
    """phase-1 statement q of an iteration whose current block has parity par.  qk: MFMAs of the NEXT block's S / dP (k-slices
    2 q, 2 q + 1; D = 64: k-slice q).  ar: 0 none, 1 plain, 2 masked arithmetic of the current block (scores 4 q ..).  tr: transpose reads of the
    current block (step q of 0 .. 3: four ds_read_b64_tr_b16; steps 4 .. 7 are read in phase 2)."""
    mf, clob = [], ["memory"]
    npar = par ^ 1
    if qk:
        s, dp = tup(c.S + 16 * npar, 16), tup(c.DP + 16 * npar, 16)
        for ks in range(q * c.KS // 4, (q + 1) * c.KS // 4):
            mf.append(f"{c.mfma} {s}, {c.frag(c.QA, ks)}, {c.frag(c.KF, ks)}, {'0' if ks == 0 else s}")
            mf.append(f"{c.mfma} {dp}, {c.frag(c.DA, ks)}, {c.frag(c.VF, ks)}, {tup(c.DL + 16 * npar, 16) if ks == 0 else dp}")
        clob += vregs(c.S + 16 * npar, 16) + vregs(c.DP + 16 * npar, 16)
    valu = arith_ops(c, par, q, ar == 2) if ar else []
    if ar:
        clob += vregs(c.T, 8) + vregs(c.P + 2 * q, 2) + vregs(c.DS + 2 * q, 2)
        if ar == 2:
            clob += ["vcc"]
    lds = []
    if tr:                                         # phase 1 carries steps 0 .. 3 (one per statement), phase 2 the other four
        steps = [q]
        if c.NST == 4 and TR64_EARLY:               # D = 64 experiment: all four steps in statements 0 and 1 (the boundary's lgkmcnt(0) finds them done)
            steps = [2 * q, 2 * q + 1] if q < 2 else []
        for st in steps:
            lds += tr_reads(c, st)
            clob += vregs(c.X + 4 * st, 4) + vregs(c.Y + 4 * st, 4)
    fill = lds + valu
    lines = deal(mf, fill)
    if qk and not fill:
        lines += ["s_nop 7", "s_nop 7"]
    ins = []
    if ar:
        ins.append('[c] "s"(c)')
        if ar == 2:
            ins += ['[lo] "v"(lo)', '[wd] "v"(wd)']
    if tr:
        ins.append('[trb] "v"(trb)')
    return emit_asm(xfilter(lines), [], ins, clob)


def gen_p2(c, q, par, mm, rm, ld, dma):
    """phase-2 statement q.  mm: dV / dK MFMAs of the current block (steps 2 q, 2 q + 1).  rm: row-major fragment reads of block
    i + 2 (k-slices 2 q, 2 q + 1 of Q and dO).  ld: statement 0 requests L' of block i + 2, statement 1 its delta (four dwordx4
    each, into the buffers of parity par; L' = LSE log2(e) comes scaled from the dQ kernel's workspace).  dma:
    statements 2 and 3 carry the eight LDS-DMA pieces of block i + 4 (image q - 2 ... : pieces 4 (q - 2) .. + 3).  All the
    requests for scalars come before all the pieces: the phase boundary's vmcnt(NP) then covers exactly the scalars."""
    mf, clob = [], ["memory"]
    if mm:
        for st in range(q * c.NST // 4, (q + 1) * c.NST // 4):
            kk, d = st // c.DB, st % c.DB
            dv, dk = c.acc(c.DV, d), c.acc(c.DK, d)
            mf.append(f"{c.mfma} {dv}, {c.frag(c.X, st, 'v')}, {c.frag(c.P, kk, 'v')}, {dv}")
            mf.append(f"{c.mfma} {dk}, {c.frag(c.Y, st, 'v')}, {c.frag(c.DS, kk, 'v')}, {dk}")
        clob += aregs(c.DV, 16 * c.DB) + aregs(c.DK, 16 * c.DB)
    fill = []
    ins = []
    pre = []
    if mm and q >= 2 and c.NST == 8:
        # (D = 128; at D = 64 all four steps are read in phase 1, behind the phase boundary's wait)
        # steps 2 q, 2 q + 1 were requested in statement q - 2; LDS returns in order, what may still be out: the reads issued after
        # them (q = 2: statement 0's row-major reads + all of statement 1's LDS reads; q = 3: statement 1's row-major reads + 2's)
        later = c.RM_SPLIT[0] + 8 + c.RM_SPLIT[1] if q == 2 else c.RM_SPLIT[1] + c.RM_SPLIT[2]
        pre.append(f"s_waitcnt lgkmcnt({min(later, 15)})")
    if mm and q < 2 and c.NST == 8:
        for st in (4 + 2 * q, 5 + 2 * q):
            fill += tr_reads(c, st)
            clob += vregs(c.X + 4 * st, 4) + vregs(c.Y + 4 * st, 4)
        ins.append('[trb] "v"(trb)')
    if rm:
        order = [(ks, t) for ks in range(c.KS) for t in (0, 1)]       # (k-slice, Q / dO)
        lo = sum(c.RM_SPLIT[:q]) if mm else len(order) // 4 * q
        n = c.RM_SPLIT[q] if mm else len(order) // 4
        for ks, t in order[lo:lo + n]:               # d = 16 ks + 8 hi ..: sub-tile ks of the lane's row group, chunk hi
            if c.D == 128:
                fill.append(f"ds_read_b128 {c.frag(c.DA if t else c.QA, ks)}, %[ra] offset:{t * c.IMG + 128 * ks}")
            else:                                    # k-slice ks = (d-slice ks >> 1, b = ks & 1): the lane base of that b, + 512 d
                fill.append(f"ds_read_b128 {c.frag(c.DA if t else c.QA, ks)}, %[ra{'1' if ks & 1 else ''}] offset:{t * c.IMG + 512 * (ks >> 1)}")
            clob += aregs((c.DA if t else c.QA) + 4 * ks, 4)
        ins += ['[ra] "v"(ra)'] + (['[ra1] "v"(ra1)'] if c.D == 64 else [])
    if ld:
        if q < 2:
            # L' of block i + 2 (its arithmetic runs in iteration i + 2); - delta of block i + 3 (the C operand of dP_{i+3}'s first
            # MFMA in iteration i + 2; the buffer of the other parity: - delta_{i+1} was consumed in this iteration's phase 1)
            base = c.LD + 16 * par if q == 0 else c.DL + 16 * (par ^ 1)
            srd = "%[lsrd]" if q == 0 else "%[dsrd]"
            so = "%[lso]" if q == 0 else "%[lso3]"
            for g in range(4):
                fill.append(f"buffer_load_dwordx4 v[{base + 4 * g}:{base + 4 * g + 3}], %[lvo], {srd}, {so} offen offset:{32 * g}")
            clob += vregs(base, 16)
            ins += ['[lsrd] "s"(lsrd)', '[lso] "s"(lso)'] if q == 0 else ['[dsrd] "s"(dsrd)', '[lso3] "s"(lso3)']
            ins += ['[lvo] "v"(lvo)']
    if dma and q >= 2:
        img = q - 2                                         # statement 2: Q, statement 3: dO; the wave's row groups 2 w, 2 w + 1
        srd = "%[qsrd]" if img == 0 else "%[gsrd]"            # (dlds = slot + PBASE[2 w]; PBASE[2 w + 1] - PBASE[2 w] = 1040)
        for half in range(c.NP // 2):                        # (D = 64: one piece per image and wave)
            fill += [f"s_add_u32 m0, %[dlds], {img * c.IMG + half * 1040}", "s_nop 0", f"buffer_load_dwordx4 %[vost{half}], {srd}, %[dso] offen lds"]
        clob += ["m0", "scc"]
        ins += ['[dlds] "s"(dlds)', '[qsrd] "s"(qsrd)' if q == 2 else '[gsrd] "s"(gsrd)', '[dso] "s"(dso)',
                '[vost0] "v"(vost0)', '[vost1] "v"(vost1)']
    lines = pre + deal(mf, fill)
    if dma and q >= 2:
        # the M0 write of a piece goes in front of the MFMA that precedes its request (the MFMA is the instruction an M0 write
        # and the request that reads it have to be apart): no s_nop -- every instruction costs this wave ~4.6 cycles of issue
        out = []
        for ln in lines:
            if ln == "s_nop 0":
                continue
            if ln.startswith("s_add_u32 m0"):
                k = max(j for j, o in enumerate(out) if o.startswith("v_mfma"))
                assert not any(o.startswith("s_add_u32 m0") or "lds" in o.split()[-1] for o in out[k:]), out[k:]
                out.insert(k, ln)
            else:
                out.append(ln)
        lines = out
    return emit_asm(xfilter(lines), [], ins, clob)


def gen_struct(c):
    name = f"Bw4Asm<{'Bf16Traits' if c.dt == 'bf16' else 'F16Traits'}, {c.D}>"
    s = f"template <> struct {name} {{\n"
    s += f"    static constexpr int NV = {c.NV}, NP = {c.NP}, SLOT = {c.SLOT}, IMG = {c.IMG}, PB1 = {c.PBASE[1]}, PB2 = {c.PBASE[2]}, PB4 = {c.PBASE[4] if len(c.PBASE) > 4 else 0}, ST_LATE = {ST_LATE};   // NV: hipcc's VGPR budget (amdgpu_num_vgpr)\n"
    s += ("    template <int Q, int PAR, int QK, int AR, int TR>\n"
          "    static __device__ __forceinline__ void p1(float c, int lo, int wd, unsigned trb) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        (void)c; (void)lo; (void)wd; (void)trb;\n")
    first = True
    for q in range(4):
        for par in range(2):
            for (qk, ar, tr) in ((1, 1, 1), (1, 2, 1), (0, 1, 1), (0, 2, 1), (1, 0, 0)):
                s += f"        {'if' if first else 'else if'} constexpr (Q == {q} && PAR == {par} && QK == {qk} && AR == {ar} && TR == {tr}) {{\n"
                s += gen_p1(c, q, par, qk, ar, tr) + "        }\n"
                first = False
    s += "        else static_assert(Q < 0, \"fa_bwd_dkv4_asm.inc: phase-1 variant not generated\");\n#endif\n    }\n"
    s += ("    template <int Q, int PAR, int MM, int RM, int LD, int DMA>\n"
          "    static __device__ __forceinline__ void p2(unsigned ra, unsigned ra1, unsigned trb, __amdgpu_buffer_rsrc_t lsrd, __amdgpu_buffer_rsrc_t dsrd, unsigned lvo,\n"
          "                                              unsigned lso, unsigned lso3, unsigned dlds, __amdgpu_buffer_rsrc_t qsrd, __amdgpu_buffer_rsrc_t gsrd, unsigned dso,\n"
          "                                              unsigned vost0, unsigned vost1) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)ra; (void)ra1; (void)trb; (void)lsrd; (void)dsrd; (void)lvo; (void)lso; (void)lso3; (void)dlds; (void)qsrd; (void)gsrd; (void)dso;\n"
          "        (void)vost0; (void)vost1;\n"
          "        if constexpr (DMA != 0) {\n            dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n            dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n        }\n"
          "        if constexpr (LD != 0) {\n            lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n            lso3 = (unsigned)__builtin_amdgcn_readfirstlane((int)lso3);\n        }\n")
    first = True
    for q in range(4):
        for par in range(2):
            for (mm, rm, ld, dma) in ((1, 1, 1, 1), (0, 1, 0, 0)):
                s += f"        {'if' if first else 'else if'} constexpr (Q == {q} && PAR == {par} && MM == {mm} && RM == {rm} && LD == {ld} && DMA == {dma}) {{\n"
                s += gen_p2(c, q, par, mm, rm, ld, dma) + "        }\n"
                first = False
    s += "        else static_assert(Q < 0, \"fa_bwd_dkv4_asm.inc: phase-2 variant not generated\");\n#endif\n    }\n"
    # ---- K / V fragments of the wave's 32 key rows: lane (key, hi) holds d = 16 ks + 8 hi .. + 7 (rows >= Sk read as 0)
    lines = ["s_nop 4"]
    for ks in range(c.KS):
        lines.append(f"buffer_load_dwordx4 {c.frag(c.KF, ks)}, %[vo], %[ksrd], 0 offen offset:{32 * ks}")
        lines.append(f"buffer_load_dwordx4 {c.frag(c.VF, ks)}, %[vo], %[vsrd], 0 offen offset:{32 * ks}")
    lines.append("s_waitcnt vmcnt(0)")
    s += "    static __device__ __forceinline__ void load_kv(__amdgpu_buffer_rsrc_t ksrd, __amdgpu_buffer_rsrc_t vsrd, unsigned vo) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], ['[ksrd] "s"(ksrd)', '[vsrd] "s"(vsrd)', '[vo] "v"(vo)'], ["memory"] + aregs(c.KF, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- accumulators to zero
    lines = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(32 * c.DB)]
    s += "    static __device__ __forceinline__ void zero_acc() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], aregs(0, 32 * c.DB), indent="        ")
    s += "#endif\n    }\n"
    # ---- L' / delta of a block straight into the buffers of parity PAR (stream start); SCALE: L' times log2(e) right away
    s += ("    template <int PAR, int SCALE>\n    static __device__ __forceinline__ void load_scal(__amdgpu_buffer_rsrc_t lsrd, __amdgpu_buffer_rsrc_t dsrd, unsigned lvo, unsigned lso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n")
    first = True
    for par in range(2):
        for scale in range(2):
            lines = ["s_nop 4"]
            for g in range(4):
                lines.append(f"buffer_load_dwordx4 v[{c.LD + 16 * par + 4 * g}:{c.LD + 16 * par + 4 * g + 3}], %[lvo], %[lsrd], %[lso] offen offset:{32 * g}")
                lines.append(f"buffer_load_dwordx4 v[{c.DL + 16 * par + 4 * g}:{c.DL + 16 * par + 4 * g + 3}], %[lvo], %[dsrd], %[lso] offen offset:{32 * g}")
            lines.append("s_waitcnt vmcnt(0)")
            if scale:
                for i in range(16):
                    lines.append(f"v_mul_f32 v{c.LD + 16 * par + i}, 0x3fb8aa3b, v{c.LD + 16 * par + i}")
            s += f"        {'if' if first else 'else if'} constexpr (PAR == {par} && SCALE == {scale}) {{\n"
            s += emit_asm(lines, [], ['[lsrd] "s"(lsrd)', '[dsrd] "s"(dsrd)', '[lvo] "v"(lvo)', '[lso] "s"(lso)'],
                          ["memory"] + vregs(c.LD + 16 * par, 16) + vregs(c.DL + 16 * par, 16))
            s += "        }\n"
            first = False
    s += "#endif\n    }\n"
    # ---- stream start: - delta of block 2 (the C operand of dP_2 in iteration 1) parked in X, moved to DL[0] once dP_0 is under way
    lines = ["s_nop 4"] + [f"buffer_load_dwordx4 v[{c.X + 4 * g}:{c.X + 4 * g + 3}], %[lvo], %[dsrd], %[lso] offen offset:{32 * g}" for g in range(4)]
    s += ("    static __device__ __forceinline__ void load_delta_x(__amdgpu_buffer_rsrc_t dsrd, unsigned lvo, unsigned lso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n")
    s += emit_asm(lines, [], ['[dsrd] "s"(dsrd)', '[lvo] "v"(lvo)', '[lso] "s"(lso)'], ["memory"] + vregs(c.X, 16), indent="        ")
    s += "#endif\n    }\n"
    lines = ["s_nop 7", "s_nop 7"] + [f"v_mov_b32 v{c.DL + i}, v{c.X + i}" for i in range(16)]
    s += "    static __device__ __forceinline__ void mov_delta_x() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], ["memory"] + vregs(c.DL, 16), indent="        ")
    s += "#endif\n    }\n"
    # ---- the LDS-DMA pieces of one block as a statement of its own (stream start)
    lines = ["s_nop 4"]
    for pi in range(c.NP):
        img, half = pi // (c.NP // 2), pi % (c.NP // 2)
        srd = "%[qsrd]" if img == 0 else "%[gsrd]"
        lines += [f"s_add_u32 m0, %[dlds], {img * c.IMG + half * 1040}", "s_nop 0", f"buffer_load_dwordx4 %[vost{half}], {srd}, %[dso] offen lds"]
    s += ("    static __device__ __forceinline__ void dma_block(unsigned dlds, __amdgpu_buffer_rsrc_t qsrd, __amdgpu_buffer_rsrc_t gsrd, unsigned dso,\n"
          "                                                     unsigned vost0, unsigned vost1) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n        dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n")
    s += emit_asm(lines, [], ['[dlds] "s"(dlds)', '[qsrd] "s"(qsrd)', '[gsrd] "s"(gsrd)', '[dso] "s"(dso)',
                              '[vost0] "v"(vost0)', '[vost1] "v"(vost1)'], ["memory", "m0", "scc"], indent="        ")
    s += "#endif\n    }\n"
    # ---- the 5-matmul backward (round 5): the packed dS of the current block -- the B operands of dK's MFMAs, 8 registers: lane
    # (key n, hi) holds query rows 16 kk + 4 hi + {0..3} and 16 kk + 8 + 4 hi + {0..3} of k-step kk in DS[4 kk .. 4 kk + 3] -- goes
    # to the workspace as a 2 KB unit [kk][lane][16 bytes]; fa_bwd_dqs_gfx950.hip reads it back (transposed through LDS) as the
    # B operand of dQ^T += K^T dS^T, so nobody recomputes S / dP.  Issued at the top of phase 2, in front of the iteration's other
    # VMEM requests: the phase boundary's vmcnt(NP) then covers the stores too (the counter is shared on gfx9).
    nt = " nt" if ST_NT else ""
    lines = [f"buffer_store_dwordx4 v[{c.DS}:{c.DS + 3}], %[svo], %[ssrd], %[sso] offen{nt}",
             f"buffer_store_dwordx4 v[{c.DS + 4}:{c.DS + 7}], %[svo], %[ssrd], %[sso] offen offset:1024{nt}"]
    s += ("    static __device__ __forceinline__ void store_ds(__amdgpu_buffer_rsrc_t ssrd, unsigned svo, unsigned sso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        sso = (unsigned)__builtin_amdgcn_readfirstlane((int)sso);\n")
    s += emit_asm(lines, [], ['[ssrd] "s"(ssrd)', '[svo] "v"(svo)', '[sso] "s"(sso)'], ["memory"], indent="        ")
    s += "#endif\n    }\n"
    s += "};\n\n"
    return s


# ------------------------------------------------------------------------------------------------------------------------------
# Round 6: D = 64 with TWO 32-key blocks per wave (Bw4Asm2<T>; workgroup = 4 waves x 64 keys = a 256-key KV block).
#
# Why (profiles/r5_bwd_d64_ablation.txt): at D = 64 an iteration is 16 MFMAs (512 cycles of matrix pipe) inside ~1300 cycles of in-order
# issue -- the same 16 scores per lane of arithmetic, the same 24 LDS reads, 8 scalar loads and 2 LDS-DMA pieces as at D = 128 behind
# half the MFMAs.  With two key blocks A, B per wave every fragment of the query block feeds TWO MFMAs: 32 MFMAs (1024 cycles) per
# iteration for twice the arithmetic and the SAME LDS reads, scalar loads, requests, barrier and compiler scalars.
#
#     phase 1   S'_A, S'_B of block i + 1 (8 MFMAs)  |  the arithmetic of block i, key block A  |  its 16 transpose reads
#     ---- s_waitcnt vmcnt(NP) lgkmcnt(0); s_barrier ----
#     phase 2   dP'_A of block i + 1 (4 MFMAs: they overwrite the dP_A the arithmetic has consumed -- ONE dP buffer), dV_A, dK_A of block i (8)
#               |  the arithmetic of block i, key block B;  then dP'_B (4), dV_B, dK_B (8)  |  L' of block i + 2 (behind the arithmetic that reads
#               its buffer), the dO fragments and - delta of block i + 2 (behind the last dP' MFMA: ONE delta buffer, read as their C operand),
#               the LDS-DMA requests last.  (First build, session r6_s5: all arithmetic in phase 1 -- 8 MFMAs against 900 cycles of VALU issue
#               there, 24 MFMAs against 650 in phase 2: B8 H32 S2048 694.6 -> 590.6 us; this split: see profiles/r6_bwd_d64_k2.txt.)
#
# Register map (NV = 40 like D = 128):
#     accumulator file                                     arch VGPRs
#     a[0:31] dV_A^T   a[32:63] dV_B^T                      v[40:47]   T      v[48:63]   DL (- delta, one buffer)
#     a[64:95] dK_A^T  a[96:127] dK_B^T                     v[64:95]   LD[2]  L' by block parity
#     a[128:143] K_A   a[144:159] K_B  fragments            v[96:111]  DS_A, DS_B (8 each)      v[112:127] P_A, P_B
#     a[160:175] V_A   a[176:191] V_B                       v[128:159] DP_A, DP_B (one buffer)
#     a[192:207] Q_b   a[208:223] dO_b  row-major           v[160:223] S[parity][A, B]
#                                                           v[224:239] Y   v[240:255] X   transposed fragments of Q_b / dO_b
# BW4_K2_SPLIT=1 (experiment, measured slower: profiles/r6_bwd_d64_k2.txt): key block B's arithmetic in phase 2, behind the dV_A / dK_A MFMAs, instead of
# phase 1 -- phase 1 then has 8 MFMAs for half the VALU work, phase 2 24 MFMAs for the other half
K2_SPLIT = os.environ.get("BW4_K2_SPLIT", "0") == "1"


class Cfg2:
    def __init__(self, dt):
        b = Cfg(64, dt)
        self.D, self.dt, self.RB, self.KS, self.DB, self.NST = 64, dt, 128, 4, 2, 4
        self.mfma, self.cvt = b.mfma, b.cvt
        self.PBASE, self.IMG, self.NP, self.SLOT = b.PBASE, b.IMG, b.NP, b.SLOT
        self.DV, self.DK = (0, 32), (64, 96)
        self.KF, self.VF = (128, 144), (160, 176)
        self.QA, self.DA = 192, 208
        self.X, self.Y = 240, 224
        self.DP, self.P, self.DS = (128, 144), (112, 120), (96, 104)
        self.LD, self.DL, self.T, self.NV = (64, 80), 48, 40, 40

    def S(self, par, blk):
        return 160 + 32 * par + 16 * blk

    def frag(self, base, i, file="a"):
        return f"{file}[{base + 4 * i}:{base + 4 * i + 3}]"


def arith2(c, par, blk, q, masked):
    """P / dS of scores 4 q .. 4 q + 3 of key block blk (the arithmetic of arith_ops on this map; masks per key block)"""
    S, DPr, LD = c.S(par, blk), c.DP[blk], c.LD[par]
    t = [c.T + i for i in range(8)]
    r0 = 4 * q
    sfx = "ab"[blk]
    ops = [f"v_fma_f32 v{t[i]}, v{S + r0 + i}, %[c], -v{LD + r0 + i}" for i in range(4)]
    ops += [f"v_exp_f32 v{t[i]}, v{t[i]}" for i in range(4)]
    if masked:
        tm = c.DS[blk] + r0 // 2
        for i in range(4):
            ops += [f"v_sub_u32 v{tm}, {crow(r0 + i)}, %[lo{sfx}]", f"v_cmp_gt_u32 vcc, %[wd{sfx}], v{tm}", f"v_cndmask_b32 v{t[i]}, 0, v{t[i]}, vcc"]
    ops += [f"v_mul_f32 v{t[4 + i]}, v{t[i]}, v{DPr + r0 + i}" for i in range(4)]
    for j in range(2):
        ops.append(f"{c.cvt} v{c.P[blk] + r0 // 2 + j}, v{t[2 * j]}, v{t[2 * j + 1]}")
        ops.append(f"{c.cvt} v{c.DS[blk] + r0 // 2 + j}, v{t[4 + 2 * j]}, v{t[5 + 2 * j]}")
    return ops


def gen2_p1(c, q, par, qk, ar, tr):
    """phase-1 statement q: S' of block i + 1 for BOTH key blocks (k-slice q), the arithmetic of block i for key block A only (scores 4 q ..; key
    block B's runs in phase 2, behind the dV_A / dK_A MFMAs: the VALU work is split over the two phases like the MFMAs), transpose reads of step q."""
    mf, clob = [], ["memory"]
    npar = par ^ 1
    if qk:
        for b in (0, 1):
            s = tup(c.S(npar, b), 16)
            mf.append(f"{c.mfma} {s}, {c.frag(c.QA, q)}, {c.frag(c.KF[b], q)}, {'0' if q == 0 else s}")
            clob += vregs(c.S(npar, b), 16)
    valu = []
    blocks = (0,) if K2_SPLIT else (0, 1)
    if ar:
        for b in blocks:
            valu += arith2(c, par, b, q, ar == 2)
            clob += vregs(c.P[b] + 2 * q, 2) + vregs(c.DS[b] + 2 * q, 2)
        clob += vregs(c.T, 8) + (["vcc"] if ar == 2 else [])
    lds = []
    if tr:
        lds = tr_reads(c, q)
        clob += vregs(c.X + 4 * q, 4) + vregs(c.Y + 4 * q, 4)
    fill = lds + valu
    lines = deal(mf, fill)
    if qk and not fill:
        lines += ["s_nop 7", "s_nop 7"]
    ins = []
    if ar:
        ins.append('[c] "s"(c)')
        if ar == 2:
            ins += ['[loa] "v"(loa)', '[wda] "v"(wda)'] + ([] if K2_SPLIT else ['[lob] "v"(lob)', '[wdb] "v"(wdb)'])
    if tr:
        ins.append('[trb] "v"(trb)')
    return emit_asm(xfilter(lines), [], ins, list(dict.fromkeys(clob)))


def gen2_p2(c, q, par, mm, rm, ld, dma, qk, ar):
    """phase-2 statement q.  MFMA order: dP'_A (4), dV_A / dK_A (8) -- statements 0, 1, with the arithmetic of block i for key block B as their
    filler (ar: 0 none, 1 plain, 2 masked; two groups of four scores per statement) --, then dP'_B (4: behind the arithmetic, which reads the dP_B
    they overwrite), dV_B / dK_B (8) -- statements 2, 3, with L' of block i + 2 (behind the arithmetic, which reads the buffer it lands in), the dO
    fragments of block i + 2 and - delta of block i + 2 (both behind the last dP' MFMA) and, last of all requests, the two LDS-DMA pieces."""
    L, clob = [], ["memory"]

    def dps(b):
        dp = tup(c.DP[b], 16)
        return [f"{c.mfma} {dp}, {c.frag(c.DA, ks)}, {c.frag(c.VF[b], ks)}, {tup(c.DL, 16) if ks == 0 else dp}" for ks in range(c.KS)]

    def mms(b):
        out = []
        for st in range(c.NST):
            kk, d = st // c.DB, st % c.DB
            dv = f"a[{c.DV[b] + 16 * d}:{c.DV[b] + 16 * d + 15}]"
            dk = f"a[{c.DK[b] + 16 * d}:{c.DK[b] + 16 * d + 15}]"
            out.append(f"{c.mfma} {dv}, {c.frag(c.X, st, 'v')}, {c.frag(c.P[b], kk, 'v')}, {dv}")
            out.append(f"{c.mfma} {dk}, {c.frag(c.Y, st, 'v')}, {c.frag(c.DS[b], kk, 'v')}, {dk}")
        return out

    if K2_SPLIT:
        for b in (0, 1):
            L += (dps(b) if qk else []) + (mms(b) if mm else [])
    else:
        # all of dP' first (key blocks interleaved: the chains alternate), then the dV / dK steps with both key blocks behind each fragment
        da, db_, ma, mb = dps(0), dps(1), mms(0), mms(1)
        if qk:
            L += [x for pair in zip(da, db_) for x in pair]
        if mm:
            for st in range(c.NST):
                L += [ma[2 * st], mb[2 * st], ma[2 * st + 1], mb[2 * st + 1]]
    per = (len(L) + 3) // 4
    mf = L[q * per:(q + 1) * per]
    for ln in mf:
        dst = ln.split()[1].rstrip(",")
        lo, hi = (int(x) for x in dst[2:-1].split(":"))
        clob += (aregs if dst[0] == "a" else vregs)(lo, hi - lo + 1)
    fill, ins = [], []
    if ar and q < 2 and K2_SPLIT:
        for grp in (2 * q, 2 * q + 1):
            fill += arith2(c, par, 1, grp, ar == 2)
            clob += vregs(c.P[1] + 2 * grp, 2) + vregs(c.DS[1] + 2 * grp, 2)
        clob += vregs(c.T, 8) + (["vcc"] if ar == 2 else [])
        ins.append('[c] "s"(c)')
        if ar == 2:
            ins += ['[lob] "v"(lob)', '[wdb] "v"(wdb)']
    if rm:
        if mm and not K2_SPLIT:
            # Q fragments (k-slices 2 q, 2 q + 1) in statements 0, 1; dO fragments in statements 2, 3: behind the last dP' MFMA (statement 1's
            # second), which reads the dO fragments of block i + 1 these reads overwrite
            t = 0 if q < 2 else 1
            reads = [(2 * (q & 1), t), (2 * (q & 1) + 1, t)]
        elif mm:
            # Q fragments (k-slices 2 q, 2 q + 1) in statements 0, 1; the four dO fragments in statement 3: behind the last dP' MFMA (statement
            # 2's fourth), which reads the dO fragments of block i + 1 these reads overwrite
            reads = [(2 * q, 0), (2 * q + 1, 0)] if q < 2 else ([(ks, 1) for ks in range(c.KS)] if q == 3 else [])
        else:
            order = [(ks, t) for ks in range(c.KS) for t in (0, 1)]
            reads = order[2 * q:2 * q + 2]
        for ks, t in reads:
            fill.append(f"ds_read_b128 {c.frag(c.DA if t else c.QA, ks)}, %[ra{'1' if ks & 1 else ''}] offset:{t * c.IMG + 512 * (ks >> 1)}")
            clob += aregs((c.DA if t else c.QA) + 4 * ks, 4)
        ins += ['[ra] "v"(ra)', '[ra1] "v"(ra1)']
    ldq = 2 if K2_SPLIT else 0      # L' in statement ldq, - delta in statement ldq + 1 (not split: L' may go first -- its buffer was consumed in phase 1;
    #                                 - delta behind statement 0's first two MFMAs, which read the buffer as their C operand)
    if ld and ldq <= q < ldq + 2:
        base = c.LD[par] if q == ldq else c.DL
        srd = "%[lsrd]" if q == ldq else "%[dsrd]"
        for g in range(4):
            fill.append(f"buffer_load_dwordx4 v[{base + 4 * g}:{base + 4 * g + 3}], %[lvo], {srd}, %[lso] offen offset:{32 * g}")
        clob += vregs(base, 16)
        ins += ['[lsrd] "s"(lsrd)' if q == ldq else '[dsrd] "s"(dsrd)', '[lso] "s"(lso)', '[lvo] "v"(lvo)']
    lines = deal(mf, fill)
    if mf and not fill and not mm:
        lines += ["s_nop 7", "s_nop 7"]
    if dma and q == 3:
        # the two pieces behind everything else (the boundary's vmcnt(NP) must leave exactly them out): the M0 write of a piece in front of one
        # of the statement's last two MFMAs, its request behind that MFMA (an M0 write and the request that reads it have to be an instruction apart)
        mi = [j for j, o in enumerate(lines) if o.startswith("v_mfma")]
        assert len(mi) >= 2
        tail = lines[mi[-1] + 1:]                       # fillers dealt behind the last MFMA: they go in front of it
        lines = lines[:mi[-2]] + lines[mi[-2] + 1:mi[-1]] + tail + \
            [f"s_add_u32 m0, %[dlds], 0", lines[mi[-2]], "buffer_load_dwordx4 %[vost0], %[qsrd], %[dso] offen lds",
             f"s_add_u32 m0, %[dlds], {c.IMG}", lines[mi[-1]], "buffer_load_dwordx4 %[vost0], %[gsrd], %[dso] offen lds"]
        clob += ["m0", "scc"]
        ins += ['[dlds] "s"(dlds)', '[qsrd] "s"(qsrd)', '[gsrd] "s"(gsrd)', '[dso] "s"(dso)', '[vost0] "v"(vost0)']
    return emit_asm(xfilter(lines), [], ins, list(dict.fromkeys(clob)))


def gen2_struct(c):
    name = f"Bw4Asm2<{'Bf16Traits' if c.dt == 'bf16' else 'F16Traits'}>"
    s = f"template <> struct {name} {{\n"
    s += f"    static constexpr int NV = {c.NV}, NP = {c.NP}, SLOT = {c.SLOT}, IMG = {c.IMG}, PB1 = {c.PBASE[1]}, PB2 = {c.PBASE[2]}, PB4 = 0, ST_LATE = 0, SPLIT = {int(K2_SPLIT)};   // NV: hipcc's VGPR budget (amdgpu_num_vgpr); SPLIT: key block B's arithmetic in phase 2\n"
    s += ("    template <int Q, int PAR, int QK, int AR, int TR>\n"
          "    static __device__ __forceinline__ void p1(float c, int loa, int wda, int lob, int wdb, unsigned trb) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        (void)c; (void)loa; (void)wda; (void)lob; (void)wdb; (void)trb;\n")
    first = True
    for q in range(4):
        for par in range(2):
            for (qk, ar, tr) in ((1, 1, 1), (1, 2, 1), (0, 1, 1), (0, 2, 1), (1, 0, 0)):
                s += f"        {'if' if first else 'else if'} constexpr (Q == {q} && PAR == {par} && QK == {qk} && AR == {ar} && TR == {tr}) {{\n"
                s += gen2_p1(c, q, par, qk, ar, tr) + "        }\n"
                first = False
    s += "        else static_assert(Q < 0, \"fa_bwd_dkv4_asm.inc: phase-1 variant not generated\");\n#endif\n    }\n"
    s += ("    template <int Q, int PAR, int MM, int RM, int LD, int DMA, int QK, int AR>\n"
          "    static __device__ __forceinline__ void p2(float c, int lob, int wdb, unsigned ra, unsigned ra1, __amdgpu_buffer_rsrc_t lsrd, __amdgpu_buffer_rsrc_t dsrd, unsigned lvo,\n"
          "                                              unsigned lso, unsigned dlds, __amdgpu_buffer_rsrc_t qsrd, __amdgpu_buffer_rsrc_t gsrd, unsigned dso, unsigned vost0) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)c; (void)lob; (void)wdb; (void)ra; (void)ra1; (void)lsrd; (void)dsrd; (void)lvo; (void)lso; (void)dlds; (void)qsrd; (void)gsrd; (void)dso; (void)vost0;\n"
          "        if constexpr (DMA != 0) {\n            dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n            dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n        }\n"
          "        if constexpr (LD != 0) lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n")
    first = True
    for q in range(4):
        for par in range(2):
            for (mm, rm, ld, dma, qk, ar) in ((1, 1, 1, 1, 1, 1), (1, 1, 1, 1, 1, 2), (1, 1, 1, 1, 0, 1), (1, 1, 1, 1, 0, 2), (0, 1, 0, 0, 0, 0), (0, 0, 0, 0, 1, 0)):
                s += f"        {'if' if first else 'else if'} constexpr (Q == {q} && PAR == {par} && MM == {mm} && RM == {rm} && LD == {ld} && DMA == {dma} && QK == {qk} && AR == {ar}) {{\n"
                s += gen2_p2(c, q, par, mm, rm, ld, dma, qk, ar) + "        }\n"
                first = False
    s += "        else static_assert(Q < 0, \"fa_bwd_dkv4_asm.inc: phase-2 variant not generated\");\n#endif\n    }\n"
    # ---- K / V fragments of the wave's two 32-key blocks: lane (key, hi) holds d = 16 ks + 8 hi .. + 7 of key row n0w + key (vo) and n0w + 32 + key (vo1)
    lines = ["s_nop 4"]
    for b, vo in ((0, "%[vo]"), (1, "%[vo1]")):
        for ks in range(c.KS):
            lines.append(f"buffer_load_dwordx4 {c.frag(c.KF[b], ks)}, {vo}, %[ksrd], 0 offen offset:{32 * ks}")
            lines.append(f"buffer_load_dwordx4 {c.frag(c.VF[b], ks)}, {vo}, %[vsrd], 0 offen offset:{32 * ks}")
    lines.append("s_waitcnt vmcnt(0)")
    s += "    static __device__ __forceinline__ void load_kv(__amdgpu_buffer_rsrc_t ksrd, __amdgpu_buffer_rsrc_t vsrd, unsigned vo, unsigned vo1) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], ['[ksrd] "s"(ksrd)', '[vsrd] "s"(vsrd)', '[vo] "v"(vo)', '[vo1] "v"(vo1)'], ["memory"] + aregs(c.KF[0], 64), indent="        ")
    s += "#endif\n    }\n"
    lines = [f"v_accvgpr_write_b32 a{i}, 0" for i in range(128)]
    s += "    static __device__ __forceinline__ void zero_acc() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], aregs(0, 128), indent="        ")
    s += "#endif\n    }\n"
    # ---- stream start: L' of a block into LD[PAR]; WITH_DL: its - delta into the delta buffer too
    s += ("    template <int PAR, int WITH_DL>\n    static __device__ __forceinline__ void load_scal(__amdgpu_buffer_rsrc_t lsrd, __amdgpu_buffer_rsrc_t dsrd, unsigned lvo, unsigned lso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n")
    first = True
    for par in range(2):
        for wd in range(2):
            lines = ["s_nop 4"]
            for g in range(4):
                lines.append(f"buffer_load_dwordx4 v[{c.LD[par] + 4 * g}:{c.LD[par] + 4 * g + 3}], %[lvo], %[lsrd], %[lso] offen offset:{32 * g}")
                if wd:
                    lines.append(f"buffer_load_dwordx4 v[{c.DL + 4 * g}:{c.DL + 4 * g + 3}], %[lvo], %[dsrd], %[lso] offen offset:{32 * g}")
            lines.append("s_waitcnt vmcnt(0)")
            s += f"        {'if' if first else 'else if'} constexpr (PAR == {par} && WITH_DL == {wd}) {{\n"
            s += emit_asm(lines, [], ['[lsrd] "s"(lsrd)', '[dsrd] "s"(dsrd)', '[lvo] "v"(lvo)', '[lso] "s"(lso)'],
                          ["memory"] + vregs(c.LD[par], 16) + (vregs(c.DL, 16) if wd else []))
            s += "        }\n"
            first = False
    s += "#endif\n    }\n"
    # ---- stream start: - delta of block 1 (the C operand of dP_1 in iteration 0's phase 2) parked in X, moved to DL once dP_0 is under way
    lines = ["s_nop 4"] + [f"buffer_load_dwordx4 v[{c.X + 4 * g}:{c.X + 4 * g + 3}], %[lvo], %[dsrd], %[lso] offen offset:{32 * g}" for g in range(4)]
    s += ("    static __device__ __forceinline__ void load_delta_x(__amdgpu_buffer_rsrc_t dsrd, unsigned lvo, unsigned lso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        lso = (unsigned)__builtin_amdgcn_readfirstlane((int)lso);\n")
    s += emit_asm(lines, [], ['[dsrd] "s"(dsrd)', '[lvo] "v"(lvo)', '[lso] "s"(lso)'], ["memory"] + vregs(c.X, 16), indent="        ")
    s += "#endif\n    }\n"
    lines = ["s_nop 7", "s_nop 7"] + [f"v_mov_b32 v{c.DL + i}, v{c.X + i}" for i in range(16)]
    s += "    static __device__ __forceinline__ void mov_delta_x() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], ["memory"] + vregs(c.DL, 16), indent="        ")
    s += "#endif\n    }\n"
    lines = ["s_nop 4"]
    for img in range(2):
        srd = "%[qsrd]" if img == 0 else "%[gsrd]"
        lines += [f"s_add_u32 m0, %[dlds], {img * c.IMG}", "s_nop 0", f"buffer_load_dwordx4 %[vost0], {srd}, %[dso] offen lds"]
    s += ("    static __device__ __forceinline__ void dma_block(unsigned dlds, __amdgpu_buffer_rsrc_t qsrd, __amdgpu_buffer_rsrc_t gsrd, unsigned dso, unsigned vost0) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n        dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n")
    s += emit_asm(lines, [], ['[dlds] "s"(dlds)', '[qsrd] "s"(qsrd)', '[gsrd] "s"(gsrd)', '[dso] "s"(dso)', '[vost0] "v"(vost0)'], ["memory", "m0", "scc"], indent="        ")
    s += "#endif\n    }\n"
    s += "};\n\n"
    return s


def main():
    hdr = ("// fa_bwd_dkv4_asm.inc -- GENERATED by tools/gen_bw4.py (do not edit; edit the generator and re-run it).\n"
           "// Instruction streams of the one-wave-per-SIMD dK / dV kernel: register map, pipeline and hazards in the generator's docstring.\n"
           "// Included by fa_bwd_dkv4_gfx950.hip inside namespace aule_hip::{anonymous}.\n\n"
           "template <class T, int D> struct Bw4Asm;\n"
           "template <class T> struct Bw4Asm2;   // D = 64, two 32-key blocks per wave (round 6)\n\n")
    body = ""
    for D in (128, 64):
        for dt in ("bf16", "fp16"):
            body += gen_struct(Cfg(D, dt))
    for dt in ("bf16", "fp16"):
        body += gen2_struct(Cfg2(dt))
    with open(OUT, "w") as fh:
        fh.write(hdr + body)
    for D in (128, 64):
        c = Cfg(D, "bf16")
        print(f"D={D}: NV={c.NV} T={c.T} DL={c.DL} LD={c.LD} DS={c.DS} P={c.P} DP={c.DP} S={c.S} Y={c.Y} X={c.X}; acc DV={c.DV} DK={c.DK} KF={c.KF} VF={c.VF} QA={c.QA} DA={c.DA}")
    print(f"wrote {OUT}: {len((hdr + body).splitlines())} lines; NV={c.NV} T={c.T} DL={c.DL} LD={c.LD} DS={c.DS} P={c.P} DP={c.DP} S={c.S} Y={c.Y} X={c.X}")


if __name__ == "__main__":
    main()
