#!/usr/bin/env python3
"""Writes aule-attention_amd/csrc/fa_fwd_w4_asm.inc: the hand-placed instruction streams of the 4-wave x 64-row forward
(fa_fwd_w4_gfx950.hip).  Run it after editing; the output is committed (the build does not need Python).

One wave per SIMD owns the whole 512-register file, and the tile loop names EVERY register it uses literally.  The kernel is
compiled with amdgpu_num_vgpr(NV): hipcc allocates v0 .. v(NV-1) only (addresses, loop scalars, the epilogue's temporaries) and
must never touch an accumulator register -- csrc/Makefile audits the .s for that (no scratch, no v_accvgpr_* outside the
streams).  Map for D = 128 (D = 64: the same order, half the sizes):

    accumulator file                                         arch VGPRs
    a[0:127]    O^T, block (qb, d) at a[(qb DB + d) 16 ..]   v[0:NV)      hipcc
    a[128:191]  Q fragments (qb, ks) at a[128 + (qb KS + ks) 4 ..]     then x0-x3 (softmax temporaries), lA0 lA1 lB0 lB1 (row sums),
    a[192:255]  K fragments of ONE tile, (ks, h) at          nmA nmB (- reference), -inf,
                a[192 + (2 ks + h) 4 ..]                     S[A] = xa0 xa1 (2 x 16), S[B] half 0 = yb0 (16), S[B] half 1 = yb1[2] (2 x 16, by tile
                                                             parity), P[A] = pA[2][4] (by parity, 4 registers each), P[B] = pB[4],
                                                             V^T fragments of ONE tile (sk, d) at VB0 + (sk DB + d) 4 ..

(qb = 32-row half of the wave's 64 rows: block A / B; h = 32-key half of the 64-key tile; ks = 16-wide slice of D; sk = 16-key slice.)

A tile step j (PAR = j & 1) is two phases of 4 statements; a statement is n MFMAs (n = KS in phase 1, 2 DB in phase 2) with its
fillers PLACED in the MFMA gaps (a gap hides ~5 single-issue instructions next to a 32-cycle MFMA):

    phase 1, statement Q:  S_{j+1} += K_{j+1} Q^T   |  softmax of 8 scores of S_j[B] -> P_j[B][Q]  |  2 DB transpose reads of V_j
                           (| LDS-DMA pieces of the tiles the stream requests this step)
    phase 2, statement Q:  O^T += V_j^T P_j^T       |  softmax of 8 scores of S_{j+1}[A] -> P_{j+1}[A][Q] |  KS / 2 reads of K_{j+2}

The K and V tiles sit in 3-deep LDS rings; the ring slot of a read is in its address register (the kernel adds slot x tile bytes),
so PAR only selects register copies.  S[B] half 0 is single-buffered (its old tile is consumed in statements 0-1 of phase 1, the new one is born in statements 2-3);
half 1 is consumed while its successor is being accumulated, hence the parity copies.  The part prologue is "step -1" (PAR = 1).

Hazards the strings take care of themselves (inline asm is invisible to hipcc's hazard recogniser):
  * v_exp_f32 result -> next VALU reader: at least one instruction between (trans forwarding);
  * MFMA result -> VALU reader: every S block is read at least 8 MFMAs after its last MFMA; bare MFMA statements end with
    s_nop padding;
  * s_add m0 -> buffer_load ... lds: one instruction between;
  * a descriptor fresh from v_readfirstlane -> VMEM: statements that open with a VMEM instruction open with s_nop 4;
  * LDS reads are complete (s_waitcnt lgkmcnt(0)) at the end of the last statement of phase 1 / in the kernel's step_end.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("W4_OUT", os.path.join(ROOT, "aule-attention_amd", "csrc", "fa_fwd_w4_asm.inc"))
# Timing experiments only (tools/w4_experiments.sh; results are garbage): W4_X = comma list of
#   noexp   v_exp_f32 -> v_mov_b32          nolds   no operand reads in the tile steps (stale fragments)
#   nodma   no LDS-DMA pieces in the plain step     novalu  no softmax arithmetic at all      nocvt  no packing
#   nobarrier / novmcnt   the plain step without its s_barrier / its counted wait (racy: cycles only)
XFLAGS = set(filter(None, os.environ.get("W4_X", "").split(",")))
# how x = c s - m_ref is computed (hardware finding, profiles/r3_probe_fillers.txt: 8-byte VOP3 instructions cost the in-order
# wave several times what 4-byte VOP1 / VOP2 ones do next to the MFMAs)
SCALE_FORM = os.environ.get("W4_SCALE", "fma")
# W4_PRE=1: the pre-scaled-Q form of the D = 64 streams.  Measured (profiles/r3_w4_d64_prescale.txt): +8.5 % on C5, +4..7 % on the
# other D = 64 shapes, the plain step 2371 -> 2099 cycles -- but rounding c q back to 16 bits moves every score by ~2^-9 (bf16) /
# 2^-12 (fp16) of its magnitude: LSE off by 2e-3 on N(0,1) bf16 inputs, and the 4-sigma fp16 case out of tolerance (O error
# 5.7e-2 against 6.7e-3).  Parity comes first: OFF, kept as an experiment switch (build with -DW4_NV_D64=52).
PRE_ON = os.environ.get("W4_PRE", "0") != "0"
# Order of a statement's softmax arithmetic and how the fillers are dealt to the MFMA gaps:
#   quad  two pairs of scores at a time (4 fma, 4 exp, adds, packs), fillers dealt by COUNT             (the round's first builds)
#   pipe  software-pipelined over the 8 scores (one v_exp_f32 per slot, its fma two slots ahead, its add one behind), fillers
#         dealt by ISSUE COST against what an MFMA hides: profiles/r3_probe_fillers.txt -- ~26 cycles per gap, plain VALU 4,
#         v_exp_f32 8, ds_read_b64_tr_b16 8, ds_read_b128 16, an LDS-DMA piece ~31; four v_exp_f32 in one gap (what quad
#         produces) overrun it by 10+ cycles
# W4_MSUM=1 (experiment, OFF): fp16 D = 64 -- the row sums l from the MATRIX pipe: one more MFMA per (row block, 16-key slice) next to the
# PV MFMAs, ones (32 x 16) times P^T, instead of one v_add_f32 per score (64 v_add_f32 per step go, 8 MFMAs come; l is then the sum of the
# weights AS ROUNDED to the V dtype: fine for fp16, <= 2^-11 relative on the LSE; bf16 moved it by 1.1-1.5e-3 in round 2, past its bar).  The
# idea: at D = 64 the step is bound by VALU issue (7 VALU per MFMA against the ~5 an MFMA hides) while the matrix pipe idles half the time.
# Measured (round 4, session 24, same box, A B A B; parity green, 137 forward tests): C5 (fp16 MQA S16384) 2162 / 2153 -> 2254 / 2251 us (-4.3 %),
# B8 H32 S2048 causal 179 / 176 -> 184 / 183 (-3.5 %), B4 32q/8kv S4096 305 -> 316 (-3.6 %); only launches of a few workgroup rounds gain
# (B2 H8 S1111 31.0 -> 30.1 us).  25 % more MFMAs cost more at the chip's power limit than 29 % fewer VALU instructions save: the large
# D = 64 shapes are energy-bound too (1304 W at 1.88 GHz, profiles/r4_power_trace.txt).  OFF.
MSUM_ON = os.environ.get("W4_MSUM", "0") != "0"
ORDER = os.environ.get("W4_ORDER", "pipe")
PLACE = os.environ.get("W4_PLACE", "count")    # (cost: by the probe's issue costs -- measured WORSE than the even count: profiles/r3_w4_placement_ab.txt)
CREG = "v" if "cvgpr" in XFLAGS else "s"    # register class of the scale operand (experiment)
# Scalar registers the streams name literally (round 4).  The kernel is compiled with amdgpu_num_sgpr(NS): hipcc allocates below,
# the tile steps own s[NS:NS+13] -- between two MFMA statements of a step hipcc then has NOTHING to compute: no descriptor
# rebuilds, no piece offsets, no cursor arithmetic (round 3: ~38 scalar instructions per plain step, 150-188 per generic one).
#   s[NS:NS+3] K descriptor   s[NS+4:NS+7] V descriptor   s(NS+8) byte offset of the K tile the step requests   s(NS+9) ... V tile
#   s(NS+10) piece offset (temporary)   s(NS+12) LDS address of this wave's piece 0 of K ring slot 0 (lds0 + 1024 wave)
NS = int(os.environ.get("W4_NS", "88"))
SG_KSRD, SG_VSRD, SG_KSO, SG_VSO, SG_T, SG_LDSB = NS, NS + 4, NS + 8, NS + 9, NS + 10, NS + 12
SG_ALL = [f"s{NS + i}" for i in range(14)]


class Cfg:
    def __init__(self, D, dt):
        self.D, self.dt = D, dt
        self.RB = 2 * D
        self.KS, self.DB = D // 16, D // 32
        self.KT = 64 * self.RB          # bytes of a K tile in LDS
        self.VT = 64 * self.RB
        self.QB0 = 32 * self.DB
        self.KB0 = self.QB0 + 8 * self.KS
        self.NP = self.KT // 4096       # LDS-DMA pieces per wave and tile
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
        # arch VGPR map, top down
        self.VB0 = 256 - 16 * self.DB                  # V fragments
        self.PB = self.VB0 - 16                        # pB[4]
        self.PA = self.PB - 32                         # pA[2][4]
        self.YB1 = self.PA - 32                        # yb1[2]
        self.YB0 = self.YB1 - 16
        self.XA = self.YB0 - 32                        # xa0, xa1
        self.NINF = self.XA - 1
        self.NM = self.NINF - 3                        # nmA, nmB (+ one unused: keeps the tuples above 4-aligned)
        self.L = self.NM - 4                           # lA0 lA1 lB0 lB1
        self.X = self.L - 4                            # x0 .. x3
        # softmax temporaries.  pipe order: scores k - 3 .. k + 1 are live in slot k -> five; the fifth one is the pad register of
        # the reference pair: hipcc's budget stays at NV (two registers fewer and it spills to scratch; -inf cannot become a
        # literal: v_cndmask_b32 already reads vcc over the constant bus)
        self.order = ORDER if D == 128 else "quad"     # (D = 64, same box: pipe -0.7 % on three shapes, quad stays)
        self.T = [self.X + i for i in range(4)] + ([self.NINF - 1] if self.order == "pipe" else [])
        self.NT = len(self.T)
        # D = 64 "pre" form (experiment, see PRE_ON): Q is multiplied by c = scale log2(e) once per part (rounded back to 16 bits)
        # and - m_ref enters through the C operand of the first MFMA of every score chain, so a score costs exp + add + half a
        # cvt (2.5 VALU instead of 3.5: at D = 64 the softmax, not the matrix pipe, sets the step).  Two 16-register tuples hold
        # - m_ref.
        self.pre = (D == 64) and PRE_ON
        self.NMT = self.X - 32 if self.pre else self.X  # - m_ref of block A (16 registers), block B (16)
        self.NV = self.NMT                             # hipcc's budget
        assert self.NV % 2 == 0 and self.XA % 4 == 0, (self.NV, self.XA)
        # row sums from the matrix pipe (MSUM_ON): l of block qb = any register of a[LA + 16 qb ..] (every row of ones P^T is the same),
        # the ones fragment at a[ONES ..] -- above the K fragments, free at D = 64
        self.msum = MSUM_ON and D == 64 and dt == "f16"
        self.LA = self.KB0 + 8 * self.KS
        self.ONES = self.LA + 32
        assert not self.msum or self.ONES + 4 <= 256

    def Lacc(self, qb):
        return f"a[{self.LA + 16 * qb}:{self.LA + 16 * qb + 15}]"

    def sum_mfma(self, qb, pf, first):
        """l[qb] += ones P^T of one 16-key slice (pf = the slice's packed weights: the PV MFMAs' B operand)"""
        return f"{self.mfma} {self.Lacc(qb)}, a[{self.ONES}:{self.ONES + 3}], {pf}, {'0' if first else self.Lacc(qb)}"

    def O(self, qb, d):
        b = (qb * self.DB + d) * 16
        return f"a[{b}:{b + 15}]"

    def Q(self, qb, ks):
        b = self.QB0 + (qb * self.KS + ks) * 4
        return f"a[{b}:{b + 3}]"

    def K(self, ks, h):
        b = self.KB0 + (2 * ks + h) * 4
        return f"a[{b}:{b + 3}]"

    def V(self, sk, d):
        b = self.VB0 + (sk * self.DB + d) * 4
        return f"v[{b}:{b + 3}]"

    def Vhalf(self, sk, d, i):
        b = self.VB0 + (sk * self.DB + d) * 4 + 2 * i
        return f"v[{b}:{b + 1}]"

    # S blocks: base register of (block, half, parity)
    def sA(self, h):
        return self.XA + 16 * h

    def sB(self, h, par):
        return self.YB0 if h == 0 else self.YB1 + 16 * par

    def pA(self, par, sk):
        return self.PA + 16 * par + 4 * sk

    def pB(self, sk):
        return self.PB + 4 * sk

    def l(self, qb, i):
        return self.L + 2 * qb + i

    def nm(self, qb):
        return self.NM + qb


def tup(b, n):
    return f"v[{b}:{b + n - 1}]"


def kk(h, r):
    """key index (minus 4 hi) inside the 64-key tile of accumulator register r of 32-key half h"""
    return 32 * h + (r & 3) + 8 * (r >> 2)


def softmax_ops(c, sbase, blk_h, e0, pbase, qb, masked, sub=False):
    """VALU of 8 scores (registers sbase + e0 .. + 7 of 32-key half blk_h) -> packed P at pbase..+3, row sums of block qb.
    quad: two pairs in flight at a time (x0 x1 | x2 x3).  pipe: slot k = [exp of score k-1 | add of score k-2 | pack of the pair
    that completed | scale of score k+2], so every consumer sits at least a slot behind its producer (dependent VALU issue stalls
    the in-order wave, and with it the next MFMA; v_exp needs one instruction of distance anyway) and the 8-cycle v_exp_f32 are
    spread one per slot."""
    nm, l = f"v{c.nm(qb)}", (f"v{c.l(qb, 0)}", f"v{c.l(qb, 1)}")
    sc = [f"v{sbase + e0 + i}" for i in range(8)]
    ops = []
    if c.order == "pipe":
        x = [f"v{c.T[i % c.NT]}" for i in range(8)]
        need_x = (not c.pre) or sub or masked          # does the score get a temporary before the exponential?

        def F(i):
            o = []
            if c.pre and sub:
                o.append(f"v_add_f32 {x[i]}, {nm}, {sc[i]}")
            elif not c.pre:
                if SCALE_FORM == "fma":
                    o.append(f"v_fma_f32 {x[i]}, {sc[i]}, %[c], {nm}")
                elif SCALE_FORM == "fmac":
                    o += [f"v_mov_b32 {x[i]}, {nm}", f"v_fmac_f32 {x[i]}, %[c], {sc[i]}"]
                else:
                    o += [f"v_mul_f32 {x[i]}, %[c], {sc[i]}", f"v_add_f32 {x[i]}, {nm}, {x[i]}"]
            if masked in (1, 2):
                o.append(f"v_cmp_le_i32 vcc, {kk(blk_h, e0 + i)}, %[thr]")
                o.append(f"v_cndmask_b32 {x[i]}, v{c.NINF}, {x[i] if (not c.pre or sub) else sc[i]}, vcc")
            if masked in (2, 3):     # sliding window: the key must not lie in front of the row's first visible one (2: both bounds, 3: this one only)
                o.append(f"v_cmp_ge_i32 vcc, {kk(blk_h, e0 + i)}, %[lo]")
                o.append(f"v_cndmask_b32 {x[i]}, v{c.NINF}, {x[i]}, vcc")
            return o

        def E(i):
            return [f"v_exp_f32 {x[i]}, {x[i] if need_x else sc[i]}"]

        def A(i):
            return [] if c.msum else [f"v_add_f32 {l[i & 1]}, {l[i & 1]}, {x[i]}"]

        def C(p):
            return [f"{c.cvt} v{pbase + p}, {x[2 * p]}, {x[2 * p + 1]}"]

        ops += F(0) + F(1)
        for k in range(1, 11):
            if 0 <= k - 1 < 8:
                ops += E(k - 1)
            if 0 <= k - 2 < 8:
                ops += A(k - 2)
            if k >= 3 and (k - 3) % 2 == 0 and (k - 3) // 2 < 4:    # pair p = (k - 3) / 2: the exponentials of scores 2p, 2p + 1 were
                ops += C((k - 3) // 2)                              # issued in slots 2p + 1, 2p + 2; frees their temporaries ...
            if k + 1 < 8:
                ops += F(k + 1)                                     # ... for score k + 1 (NT = 5 temporaries: i mod 5)
    else:
        for p0 in (0, 2):
            x = {p0: (f"v{c.X}", f"v{c.X + 1}"), p0 + 1: (f"v{c.X + 2}", f"v{c.X + 3}")}
            scp = {p: (sc[2 * p], sc[2 * p + 1]) for p in (p0, p0 + 1)}
            src = x        # what the exponential reads
            for p in (p0, p0 + 1):
                for i in range(2):
                    if c.pre and sub:                # scores of a bare QK^T (tile 0): already in units of c, - m_ref still to come
                        ops.append(f"v_add_f32 {x[p][i]}, {nm}, {scp[p][i]}")
                    elif c.pre:                      # the MFMA chain started from - m_ref: nothing to do
                        if not masked:
                            src = scp
                    elif SCALE_FORM == "fma":          # one VOP3 (8 bytes)
                        ops.append(f"v_fma_f32 {x[p][i]}, {scp[p][i]}, %[c], {nm}")
                    elif SCALE_FORM == "fmac":       # two 4-byte instructions: x = - m_ref ; x += c s
                        ops.append(f"v_mov_b32 {x[p][i]}, {nm}")
                        ops.append(f"v_fmac_f32 {x[p][i]}, %[c], {scp[p][i]}")
                    elif SCALE_FORM == "mulsub":
                        ops.append(f"v_mul_f32 {x[p][i]}, %[c], {scp[p][i]}")
                        ops.append(f"v_add_f32 {x[p][i]}, {nm}, {x[p][i]}")
            if masked:
                for p in (p0, p0 + 1):
                    for i in range(2):
                        if masked in (1, 2):
                            ops.append(f"v_cmp_le_i32 vcc, {kk(blk_h, e0 + 2 * p + i)}, %[thr]")
                            ops.append(f"v_cndmask_b32 {x[p][i]}, v{c.NINF}, {(x if not (c.pre and not sub) else scp)[p][i]}, vcc")
                        if masked in (2, 3):
                            ops.append(f"v_cmp_ge_i32 vcc, {kk(blk_h, e0 + 2 * p + i)}, %[lo]")
                            ops.append(f"v_cndmask_b32 {x[p][i]}, v{c.NINF}, {x[p][i]}, vcc")
            for p in (p0, p0 + 1):
                ops.append(f"v_exp_f32 {x[p][0]}, {src[p][0]}")
                ops.append(f"v_exp_f32 {x[p][1]}, {src[p][1]}")
            for p in (p0, p0 + 1):
                if not c.msum:
                    ops.append(f"v_add_f32 {l[0]}, {l[0]}, {x[p][0]}")
                    ops.append(f"v_add_f32 {l[1]}, {l[1]}, {x[p][1]}")
                ops.append(f"{c.cvt} v{pbase + p}, {x[p][0]}, {x[p][1]}")
    if "noexp" in XFLAGS:
        ops = [o.replace("v_exp_f32", "v_mov_b32") for o in ops]
    if "dropexp" in XFLAGS:
        ops = [o for o in ops if "v_exp_f32" not in o]
    if "dropfma" in XFLAGS:
        ops = [o for o in ops if "v_fma_f32" not in o]
    if "dropadd" in XFLAGS:
        ops = [o for o in ops if "v_add_f32" not in o]
    if "nocvt" in XFLAGS:
        ops = [o for o in ops if "v_cvt_pk" not in o]
    if "novalu" in XFLAGS:
        ops = []
    return ops


def issue_cost(op):
    """cycles of the wave's issue an instruction takes next to an MFMA (profiles/r3_probe_fillers.txt)"""
    m = op.split()[0]
    if m == "v_exp_f32":
        return 8
    if m.startswith("ds_read_b64_tr"):
        return 8
    if m == "ds_read_b128":
        return 16
    if m.startswith("buffer_load"):
        return 24
    return 4


def place(mfmas, lds, valu, dma, lds_per_gap, dma_gap, lds_first_gap=0, after_gap=None):
    """Interleave: after MFMA g come that gap's fillers, in program order.  LDS reads go to the earliest gaps (lds_per_gap each);
    the DMA piece (s_add m0 / one VALU / buffer_load ... lds -- a ~40-cycle issue by itself) gets gap dma_gap nearly to itself;
    the VALU fill the other gaps to an even share.  Without MFMAs the fillers are emitted in order."""
    n = len(mfmas)
    out = []
    if n == 0:
        out += lds
        v = list(valu)
        for piece in dma:
            out.append(piece[0])
            out.append(v.pop(0) if v else "s_nop 0")
            out.append(piece[1])
        out += v
        return out
    assert len(dma) <= 1
    lds, valu = list(lds), list(valu)
    if dma and not valu:
        valu = ["s_nop 0"]      # (the instruction between the M0 write and the request)
    if PLACE == "cost":
        return place_by_cost(mfmas, lds, valu, dma, dma_gap, lds_first_gap, after_gap, lds_per_gap)
    nl = [0] * n
    rest = len(lds)
    for g in range(lds_first_gap, n):
        if dma and g == dma_gap:
            continue
        nl[g] = min(lds_per_gap, rest)
        rest -= nl[g]
    assert rest == 0
    normal = [g for g in range(n) if not (dma and g == dma_gap)]
    nv = [0] * n
    left = len(valu) - (1 if dma else 0)
    total = left + LDSW * sum(nl)
    for i, g in enumerate(normal):      # even share of (LDS + VALU) over the normal gaps, the remainder to the late ones
        share = total * (i + 1) // len(normal) - total * i // len(normal)
        nv[g] = max(0, share - LDSW * nl[g])
    # rounding: hand what is left over to (or take the excess from) the late gaps
    diff = left - sum(nv)
    g = len(normal) - 1
    assert diff >= 0 or sum(nv) >= -diff, (diff, nv)
    while diff != 0:
        if diff > 0:
            nv[normal[g]] += 1; diff -= 1
        elif nv[normal[g]] > 0:
            nv[normal[g]] -= 1; diff += 1
        g = g - 1 if g > 0 else len(normal) - 1
    if dma:
        nv[dma_gap] = 1
    for g in range(n):
        out.append(mfmas[g])
        for _ in range(nl[g]):
            out.append(lds.pop(0))
        if dma and g == dma_gap:
            out.append(dma[0][0]); out.append(valu.pop(0)); out.append(dma[0][1])
        else:
            for _ in range(nv[g]):
                out.append(valu.pop(0))
        if after_gap is not None and g == after_gap[0]:
            out += after_gap[1]
    assert not lds and not valu, (len(lds), len(valu))
    return out


GAP_BUDGET = int(os.environ.get("W4_GAP", "26"))    # cycles of other issue an MFMA hides
LDS_SPREAD = os.environ.get("W4_LDS", "early") == "spread"
# count placer knobs (A/B experiments; defaults = what ships): LDS reads per gap in phase 1 / phase 2, how far from the statement's
# end the DMA gap sits, the weight of an LDS read in the per-gap count
LDSPG1 = int(os.environ.get("W4_LDSPG1", "2"))
LDSPG2 = int(os.environ.get("W4_LDSPG2", "1"))
DMAGAP = int(os.environ.get("W4_DMAGAP", "1"))
LDSW = int(os.environ.get("W4_LDSW", "1"))


def place_by_cost(mfmas, lds, valu, dma, dma_gap, lds_first_gap, after_gap, lds_per_gap=1):
    """Deal the fillers to the MFMA gaps by issue cost (issue_cost): the LDS reads spread evenly over the gaps they may use, the
    DMA piece in its own gap with the one VALU its M0 write needs, the VALU -- in program order -- filling every gap up to the
    common level that makes the whole statement fit (at least GAP_BUDGET).  The last gap takes what is left."""
    n = len(mfmas)
    glds = [[] for _ in range(n)]
    elig = [g for g in range(lds_first_gap, n) if not (dma and g == dma_gap)]
    if LDS_SPREAD:
        for i, op in enumerate(lds):                   # evenly: read i goes to eligible gap floor(i * len / count)
            glds[elig[i * len(elig) // len(lds)]].append(op)
    else:                                              # early, lds_per_gap each: a read issued late in the statement has its latency
        for i, op in enumerate(lds):                   # exposed at the statement's (or the next step's) lgkmcnt(0)
            glds[elig[min(i // lds_per_gap, len(elig) - 1)]].append(op)
    fixed = [sum(issue_cost(o) for o in glds[g]) for g in range(n)]
    if dma:
        fixed[dma_gap] += 4 + 4 + 24               # s_add m0, the VALU behind it, the request
    total = sum(fixed) + sum(issue_cost(o) for o in valu) - (4 if dma else 0)
    level = max(GAP_BUDGET, -(-total // n))
    gv = [[] for _ in range(n)]
    v = list(valu)
    if dma:
        pass
    for g in range(n):
        if dma and g == dma_gap:
            gv[g].append(v.pop(0) if v else "s_nop 0")
            continue
        acc = fixed[g]
        while v and (acc + issue_cost(v[0]) <= level or g == n - 1):
            acc += issue_cost(v[0])
            gv[g].append(v.pop(0))
    gv[n - 1] += v      # (program order is kept: what is left goes behind everything else, also behind a DMA piece in the last gap)
    out = []
    for g in range(n):
        out.append(mfmas[g])
        out += glds[g]
        if dma and g == dma_gap:
            out += [dma[0][0], gv[g][0], dma[0][1]] + gv[g][1:]
        else:
            out += gv[g]
        if after_gap is not None and g == after_gap[0]:
            out += after_gap[1]
    return out


def emit_asm(lines, outs, ins, clobbers, indent="            "):
    if not lines:
        lines = ["s_nop 0"]
    body = "\n".join(f'{indent}    "{l}\\n\\t"' for l in lines)
    cl = ", ".join(f'"{x}"' for x in clobbers)
    return f"{indent}asm volatile(\n{body}\n{indent}    : {', '.join(outs)}\n{indent}    : {', '.join(ins)}\n{indent}    : {cl});\n"


def vregs(b, n):
    return [f"v{b + i}" for i in range(n)]


def aregs(b, n):
    return [f"a{b + i}" for i in range(n)]


def dma_piece(c, Q):
    """LDS-DMA piece a statement carries (K in phase 1, V in phase 2): piece Q of 4, or piece Q / 2 in statements 0 and 2 of 2."""
    if c.NP == 4:
        return Q
    if c.NP == 2 and Q in (0, 2):
        return Q // 2
    return None


def lit_piece(c, which, slot, piece):
    """One LDS-DMA piece of a tile with every scalar operand a literal register (dma >= 2): which = 'k' / 'v', slot = ring slot the
    tile goes to.  The first string goes in front of the VALU that separates the M0 write from the request."""
    lds = (0 if which == "k" else 3 * c.KT) + slot * c.KT + piece * 4096
    so, srd = (SG_KSO, SG_KSRD) if which == "k" else (SG_VSO, SG_VSRD)
    head = f"s_add_u32 m0, s{SG_LDSB}, {lds}"
    if piece:
        head += f"\\n\\ts_add_u32 s{SG_T}, s{so}, {piece * 4096}"
    return (head, f"buffer_load_dwordx4 %[vo], s[{srd}:{srd + 3}], s{SG_T if piece else so} offen lds")


def gen_p1(c, Q, par, qk, sm, vr, dma, sl=0):
    """phase-1 statement of step PAR.  qk: MFMAs of S_{j+1}.  sm: 0 none, 1 plain, 2 masked (S_j[B]); 3 / 4: the same for the S_0 of
    the part prologue's bare QK^T (the pre form still has to subtract the reference there; otherwise identical to 1 / 2).  vr: V_j
    reads (ring slot in the address register).  dma: this statement's piece of the K tile the step requests."""
    qb, KS, DB = Q >> 1, c.KS, c.DB
    mf, clob = [], ["memory"]
    # sliding window (round 6): 5 = the masked form with a lower bound too (%[lo]), 6 = the lower bound only (a tile that crosses the window's
    # left edge and not the diagonal); tile 0 of a part takes them as they are (no pre form)
    mk = {5: 2, 6: 3}.get(sm, 1)
    if mk != 1:
        assert not c.pre
        sm = 2
    sub = sm >= 3
    if sub:
        sm -= 2
    if qk:
        acc = [c.sA(0), c.sA(1)] if qb == 0 else [c.sB(0, par ^ 1), c.sB(1, par ^ 1)]
        # pre form: every chain but the bare ones (prologue, exact-maximum pass: sm == 0) starts from - m_ref of its block
        c0 = tup(c.NMT + 16 * qb, 16) if (c.pre and sm) else "0"
        for t in range(KS // 2):
            ks = (Q & 1) * (KS // 2) + t
            for h in range(2):
                a = tup(acc[h], 16)
                mf.append(f"{c.mfma} {a}, {c.K(ks, h)}, {c.Q(qb, ks)}, {c0 if ks == 0 else a}")
        clob += vregs(acc[0], 16) + vregs(acc[1], 16)
    lds = []
    if vr:
        sk = Q
        for d in range(DB):
            for i in range(2):
                off = sl * c.VT + ((4 * sk) * (c.D // 16) + 2 * d) * 128 + i * 2 * (c.D // 16) * 128
                assert off < 65536
                lds.append(f"ds_read_b64_tr_b16 {c.Vhalf(sk, d, i)}, %[va] offset:{off}")
        clob += vregs(c.VB0 + Q * DB * 4, DB * 4)
    valu = []
    if sm:
        h = Q >> 1
        valu = softmax_ops(c, c.sB(h, par), h, 8 * (Q & 1), c.pB(Q), 1, mk if sm == 2 else 0, sub)
        clob += [f'v{t}' for t in c.T] + vregs(c.pB(Q), 4) + vregs(c.l(1, 0), 2)
        if sm == 2:
            clob.append("vcc")
    pieces = []
    if dma and dma_piece(c, Q) is not None:
        if dma >= 2:      # K_{j+4} goes to ring slot (sl + 1) mod 3
            pieces.append(lit_piece(c, "k", (sl + 1) % 3, dma_piece(c, Q)))
            clob += ["m0", "scc", f"s{SG_T}"]
        else:
            pieces.append((f"s_add_u32 m0, %[lds], {dma_piece(c, Q) * 4096}", "buffer_load_dwordx4 %[vo], %[srd], %[so] offen lds"))
            clob += ["m0", "scc"]
    if "nolds" in XFLAGS:
        lds = []
    if "nodma" in XFLAGS:
        pieces = []
    # the step's tile barrier (statement 0 of the embedded-request forms).  dma 3: first step of a part -- the prologue's
    # vmcnt(0) left nothing of this step's tiles in flight: the barrier alone (every wave has read K_0 and K_1)
    tile_wait = [x for x in ((f"s_waitcnt vmcnt({2 * c.NP})",) if dma != 3 else ()) + ("s_barrier",)
                 if not (("nobarrier" in XFLAGS and x == "s_barrier") or ("novmcnt" in XFLAGS and "vmcnt" in x))] or ["s_nop 0"]
    pre_pieces = []
    if dma == 3:
        # step 0 also asks for K_3 (ring slot 0: where K_0 was -- every wave has read it by this statement's barrier): four pieces in
        # the gaps right behind the barrier, OLDER than the step's own request, so step 1's counted wait (everything but the newest
        # 2 NP) covers them.  Constant offset: tile 3 of the part, whatever the cursor says (round 3: four bare pieces after step 0's
        # barrier; this round's first builds: in the prologue, behind an extra barrier).
        for pi in range(c.NP):
            pre_pieces.append((f"s_add_u32 m0, s{SG_LDSB}, {pi * 4096}\\n\\ts_mov_b32 s{SG_T}, {3 * c.KT + pi * 4096}",
                               f"buffer_load_dwordx4 %[vo], s[{SG_KSRD}:{SG_KSRD + 3}], s{SG_T} offen lds"))
    if dma and Q == 0 and not mf:
        # the wave's last tile (no S_{j+1}): nothing to hide the barrier behind
        lines = ["s_waitcnt lgkmcnt(0)"] + tile_wait + place(mf, lds, valu, pieces, 2, 0)
    elif dma and Q == 0:
        # the plain step's tile barrier: the K fragments this statement's MFMAs read were requested from LDS in the previous
        # step's phase 2 (lgkmcnt(0) first); the first two MFMAs and their softmax fillers need nothing the barrier guards, so the
        # wait for the tiles requested two steps ago (everything but the previous step's 2 NP pieces) and the barrier sit
        # behind them -- the matrix pipe keeps running while the workgroup meets.  V reads and requests come after it.
        lines = ["s_waitcnt lgkmcnt(0)"] + place(mf, lds, valu, pieces, 2, len(mf) - 2 if len(mf) >= 8 else len(mf) - 1, 2 if len(mf) >= 8 else 1,
                                               (1 if len(mf) >= 8 else 0, tile_wait))
        if pre_pieces:
            # one K_3 piece behind each of the MFMAs that follow the barrier (its M0 write, one filler of that gap, the request)
            out, k, seen_barrier, armed = [], 0, False, False
            for ln in lines:
                out.append(ln)
                if ln == "s_barrier":
                    seen_barrier = True
                elif seen_barrier and ln.startswith("v_mfma") and k < len(pre_pieces):
                    armed = True
                elif armed and not ln.startswith(("ds_read", "s_", "buffer_")):
                    # (ln is a VALU filler: M0 write in front of it, request behind it)
                    out.insert(len(out) - 1, pre_pieces[k][0])
                    out.append(pre_pieces[k][1])
                    k += 1
                    armed = False
            assert k == len(pre_pieces), (k, len(pre_pieces))
            lines = out
    else:
        lines = place(mf, lds, valu, pieces, LDSPG1, len(mf) - DMAGAP if len(mf) >= 8 else len(mf) - 1)
    if vr and Q == 3:
        lines.append("s_waitcnt lgkmcnt(0)")
    if qk and not sm:
        lines += ["s_nop 7", "s_nop 7"]      # bare MFMAs: results are read by whatever comes next
    ins = []
    if sm:
        if not c.pre:
            ins.append(f'[c] "{CREG}"(c)')
        if sm == 2 and mk != 3:
            ins.append('[thr] "v"(thr)')
        if mk != 1:
            ins.append('[lo] "v"(lo)')
    if vr:
        ins.append('[va] "v"(va)')
    if pieces:
        ins += ['[vo] "v"(dvo)'] if dma >= 2 else ['[lds] "s"(dlds)', '[srd] "s"(dsrd)', '[so] "s"(dso)', '[vo] "v"(dvo)']
    return emit_asm(lines, [], ins, clob)


def gen_p2(c, Q, par, pv, sm, kr, dma, sl=0):
    """phase-2 statement of step PAR.  pv: 0 none, 1 accumulate, 2 first tile of a part (C = 0 for sk = 0).  sm: softmax of S_{j+1}[A]
    -> pA[PAR ^ 1][Q].  kr: reads of K_{j+2} (ring slot in the address registers).  dma: this statement's piece of the V tile."""
    qb, KS, DB = Q >> 1, c.KS, c.DB
    mf, clob = [], ["memory"]
    mk = {5: 2, 6: 3}.get(sm, 1)           # (as in gen_p1)
    if mk != 1:
        assert not c.pre
        sm = 2
    if pv:
        for t in range(2):
            sk = 2 * (Q & 1) + t
            pf = tup(c.pA(par, sk), 4) if qb == 0 else tup(c.pB(sk), 4)
            for d in range(DB):
                o = c.O(qb, d)
                mf.append(f"{c.mfma} {o}, {c.V(sk, d)}, {pf}, {'0' if (pv == 2 and sk == 0) else o}")
            if c.msum:
                mf.append(c.sum_mfma(qb, pf, pv == 2 and sk == 0))
        for d in range(DB):
            clob += aregs((qb * DB + d) * 16, 16)
        if c.msum:
            clob += aregs(c.LA + 16 * qb, 16)
    lds = []
    qloads = []
    if kr == 2:
        # the wave's LAST tile (no K_{j+2} to read, the Q fragments dead since the previous step's QK^T): the next part's Q rows instead --
        # 2 KS row-strided buffer loads straight into the accumulator file, KS / 2 per statement, in the gaps of the PV MFMAs.  (They used to
        # be issued behind the wave's last step, with nothing to hide behind: 1600-2300 cycles of the workgroup's critical path per part,
        # profiles/r4_w4_embedded_requests.txt.)  Block Q >> 1, k-slices KS / 2 (Q & 1) ..; %[ka0] / %[ka1]: the lane's row offsets of the two
        # blocks, %[srd]: the next part's Q rows.
        for t in range(KS // 2):
            ks = (Q & 1) * (KS // 2) + t
            qloads.append(f"buffer_load_dwordx4 {c.Q(qb, ks)}, %[ka{qb}], %[srd], 0 offen offset:{32 * ks}")
        clob += aregs(c.QB0 + (qb * KS + (Q & 1) * (KS // 2)) * 4, 4 * (KS // 2))
    if kr == 1:
        n = KS // 4
        for t in range(n):
            ks = Q * n + t
            for h in range(2):
                lds.append(f"ds_read_b128 {c.K(ks, h)}, %[ka{t}] offset:{sl * c.KT + h * 32 * c.RB}")
            clob += aregs(c.KB0 + 2 * ks * 4, 8)
    valu = []
    if sm:
        h = Q >> 1
        valu = softmax_ops(c, c.sA(h), h, 8 * (Q & 1), c.pA(par ^ 1, Q), 0, mk if sm == 2 else 0, pv == 0)
        clob += [f'v{t}' for t in c.T] + vregs(c.pA(par ^ 1, Q), 4) + vregs(c.l(0, 0), 2)
        if sm == 2:
            clob.append("vcc")
    pieces = []
    if dma and dma_piece(c, Q) is not None:
        if dma >= 2:      # V_{j+2} goes to ring slot sl (= the slot K_{j+2} is read from: (step position + 2) mod 3)
            pieces.append(lit_piece(c, "v", sl, dma_piece(c, Q)))
            clob += ["m0", "scc", f"s{SG_T}"]
        else:
            pieces.append((f"s_add_u32 m0, %[lds], {dma_piece(c, Q) * 4096}", "buffer_load_dwordx4 %[vo], %[srd], %[so] offen lds"))
            clob += ["m0", "scc"]
    if "nolds" in XFLAGS:
        lds = []
    if "nodma" in XFLAGS:
        pieces = []
    lines = place(mf, lds, valu, pieces, LDSPG2, len(mf) - DMAGAP if len(mf) >= 8 else len(mf) - 1)
    if qloads:
        # one load behind each of the statement's first MFMAs (the DMA piece sits in a late gap)
        out, k = [], 0
        if Q == 0:
            out.append("s_nop 4")          # (a descriptor fresh from the scalar unit -> VMEM)
        for ln in lines:
            out.append(ln)
            if ln.startswith("v_mfma") and k < len(qloads):
                out.append(qloads[k])
                k += 1
        assert k == len(qloads)
        lines = out
    if dma >= 2 and Q == 3:
        # the request cursors move on with the step (the kernel overrides them where a part ends: fa_fwd_w4_gfx950.hip, fix_cursors)
        lines += [f"s_add_u32 s{SG_KSO}, s{SG_KSO}, {c.KT}", f"s_add_u32 s{SG_VSO}, s{SG_VSO}, {c.VT}"]
        clob += [f"s{SG_KSO}", f"s{SG_VSO}", "scc"]
    ins = []
    if sm:
        if not c.pre:
            ins.append(f'[c] "{CREG}"(c)')
        if sm == 2 and mk != 3:
            ins.append('[thr] "v"(thr)')
        if mk != 1:
            ins.append('[lo] "v"(lo)')
    if kr == 1:
        ins += [f'[ka{t}] "v"(ka{t})' for t in range(KS // 4)]
    if kr == 2:
        ins += [f'[ka{qb}] "v"(ka{qb})', '[srd] "s"(dsrd)']
    if pieces:
        ins += ['[vo] "v"(dvo)'] if dma >= 2 else ['[lds] "s"(dlds)', '[srd] "s"(dsrd)', '[so] "s"(dso)', '[vo] "v"(dvo)']
    return emit_asm(lines, [], ins, clob)


def deal_even(mfmas, fillers):
    """MFMA g, then an even share of the fillers (program order kept)."""
    n, out, k = len(mfmas), [], 0
    for g in range(n):
        out.append(mfmas[g])
        take = len(fillers) * (g + 1) // n - len(fillers) * g // n
        out += fillers[k:k + take]
        k += take
    return out


def gen_seam(c, Q):
    """Statement Q of the SEAM (round 4): the next part's bare QK^T of tile 0 (the MFMAs of gen_p1(c, Q, 1, 1, 0, 0, 0)) with the pack of
    the FINISHED part's O^T as fillers -- block Q >> 1, d-blocks DB / 2 (Q & 1) .. : accumulator -> register, times 1 / l, rounded, written
    transposed into the wave's slab row of the lane (the epilogue's w4_pack_block, instruction for instruction, so the bytes are the same).
    The epilogue's pack ran with no MFMA around at a lone wave's ~8 cycles per instruction (profiles/r3_w4_seam_experiments.txt); here it
    rides in the gaps of MFMAs that have no other fillers.  Temporaries: the weight registers pA (dead between two parts; the rope tables
    that land there are consumed before this statement)."""
    qb, KS, DB = Q >> 1, c.KS, c.DB
    acc = [c.sA(0), c.sA(1)] if qb == 0 else [c.sB(0, 0), c.sB(1, 0)]
    mf = []
    for t in range(KS // 2):
        ks = (Q & 1) * (KS // 2) + t
        for h in range(2):
            a = tup(acc[h], 16)
            mf.append(f"{c.mfma} {a}, {c.K(ks, h)}, {c.Q(qb, ks)}, {'0' if ks == 0 else a}")
    nd = DB // 2
    fill = []
    for j in range(nd):
        d = (Q & 1) * nd + j
        n0 = (qb * DB + d) * 16
        tb = c.PA + 16 * j
        for i in range(16):
            fill.append(f"v_accvgpr_read_b32 v{tb + i}, a{n0 + i}")
        for g4 in range(4):
            for i in range(4):
                fill.append(f"v_mul_f32 v{tb + 4 * g4 + i}, v{tb + 4 * g4 + i}, %[inv]")
            fill.append(f"{c.cvt} v{tb + 4 * g4}, v{tb + 4 * g4}, v{tb + 4 * g4 + 1}")
            fill.append(f"{c.cvt} v{tb + 4 * g4 + 1}, v{tb + 4 * g4 + 2}, v{tb + 4 * g4 + 3}")
            fill.append(f"ds_write_b64 %[dst], v[{tb + 4 * g4}:{tb + 4 * g4 + 1}] offset:{(32 * d + 8 * g4) * 2}")
    lines = deal_even(mf, fill)
    if Q == 3:
        lines += ["s_nop 7", "s_nop 7"]      # the last S block is read by the row maximum that follows
    clob = ["memory"] + vregs(acc[0], 16) + vregs(acc[1], 16) + vregs(c.PA, 16 * nd)
    return emit_asm(lines, [], ['[inv] "v"(inv)', '[dst] "v"(dst)'], clob)


def gen_diag(c, I, par, sl, ql):
    """Statement I (0 .. 5) of the wave's LAST tile in the overlapped form (round 4): the tile has no S_{j+1}, so its phase 1 -- the masked
    softmax of S_j[B], 176 VALU and 32 transpose reads -- ran with no MFMA around (1400-1900 cycles at a lone wave's issue rate), then
    phase 2 the 32 PV MFMAs nearly bare.  Block A's half of the PV needs nothing of this tile's arithmetic (P_j[A] is a step old), only
    V_j's fragments: statements 1-3 carry O^T[A] += V_j^T P_j[A]^T of key slice I - 1 (read one statement earlier) next to the softmax of
    slice I; statement 4 the last slice of block A and the first two of block B, statement 5 the rest.  Same accumulation order per
    accumulator as the two-phase form (key slices 0 .. 3), so the bytes are the same.
    sl: stream position mod 3 (V_j in slot sl, K request to slot sl + 1, V request to slot sl + 2).  ql: statements 4 / 5 also carry the
    next part's Q rows (see gen_p2, kr = 2)."""
    KS, DB = c.KS, c.DB
    clob = ["memory"]
    mf, lds, valu, pieces, qloads, pre, ins = [], [], [], [], [], [], []
    def pv(qb, sk):
        pf = tup(c.pA(par, sk), 4) if qb == 0 else tup(c.pB(sk), 4)
        out = []
        for d in range(DB):
            o = c.O(qb, d)
            out.append(f"{c.mfma} {o}, {c.V(sk, d)}, {pf}, {o}")
        if c.msum:
            out.append(c.sum_mfma(qb, pf, False))
        return out
    if I < 4:
        Q = I
        if I == 0:
            pre = ["s_waitcnt lgkmcnt(0)", f"s_waitcnt vmcnt({2 * c.NP})", "s_barrier"]
        else:
            pre = ["s_waitcnt lgkmcnt(0)"]          # V_j's slice I - 1 (requested a statement ago) is back
            mf = pv(0, I - 1)
            for d in range(DB):
                clob += aregs(d * 16, 16)
            if c.msum:
                clob += aregs(c.LA, 16)
        sk = Q
        for d in range(DB):
            for i in range(2):
                off = sl * c.VT + ((4 * sk) * (c.D // 16) + 2 * d) * 128 + i * 2 * (c.D // 16) * 128
                lds.append(f"ds_read_b64_tr_b16 {c.Vhalf(sk, d, i)}, %[va] offset:{off}")
        clob += vregs(c.VB0 + Q * DB * 4, DB * 4)
        h = Q >> 1
        valu = softmax_ops(c, c.sB(h, par), h, 8 * (Q & 1), c.pB(Q), 1, True, False)
        clob += [f'v{t}' for t in c.T] + vregs(c.pB(Q), 4) + vregs(c.l(1, 0), 2) + ["vcc"]
        if dma_piece(c, Q) is not None:
            pieces.append(lit_piece(c, "k", (sl + 1) % 3, dma_piece(c, Q)))
            clob += ["m0", "scc", f"s{SG_T}"]
        ins = ['[c] "s"(c)', '[thr] "v"(thr)', '[va] "v"(va)'] + (['[vo] "v"(dvo)'] if pieces else [])
        # few MFMAs, many fillers: the reads behind the first MFMA (their data is wanted a statement later), the VALU in even
        # shares over the gaps, the request (M0 write / one VALU / load) in the last one
        body = list(valu)
        if pieces:
            k = len(body) - 6          # near the end: the request is the slowest filler to issue
            body = body[:k] + [pieces[0][0], body[k], pieces[0][1]] + body[k + 1:]
        fill = lds + body
        lines = pre + (deal_even(mf, fill) if mf else fill)
        if I == 3:
            lines.append("s_waitcnt lgkmcnt(0)")
    else:
        J = I - 4
        if J == 0:
            mf = pv(0, 3) + pv(1, 0) + pv(1, 1)
        else:
            mf = pv(1, 2) + pv(1, 3)
        for d in range(2 * DB):
            clob += aregs(d * 16, 16)
        if c.msum:
            clob += aregs(c.LA, 32)
        npc = c.NP // 2
        for pi in range(J * npc, (J + 1) * npc):
            pieces.append(lit_piece(c, "v", (sl + 2) % 3, pi))
        clob += ["m0", "scc", f"s{SG_T}"]
        if ql:
            for t in range(KS):
                ks = t
                qb = J
                qloads.append(f"buffer_load_dwordx4 {c.Q(qb, ks)}, %[ka{qb}], %[srd], 0 offen offset:{32 * ks}")
            clob += aregs(c.QB0 + J * KS * 4, 4 * KS)
            ins += [f'[ka{J}] "v"(ka{J})', '[srd] "s"(dsrd)']
        ins += ['[vo] "v"(dvo)']
        out, k = (["s_nop 4"] if (ql and J == 0) else []), 0
        for g, m in enumerate(mf):
            out.append(m)
            if k < len(qloads):
                out.append(qloads[k]); k += 1
        assert k == len(qloads) or not ql, (k, len(qloads))
        # the V pieces behind the last MFMAs (M0 write, one instruction, the request)
        for head, load in pieces:
            out += [head, "s_nop 0", load]
        lines = out
        if J == 1:
            lines += [f"s_add_u32 s{SG_KSO}, s{SG_KSO}, {c.KT}", f"s_add_u32 s{SG_VSO}, s{SG_VSO}, {c.VT}"]
            clob += [f"s{SG_KSO}", f"s{SG_VSO}"]
    return emit_asm(lines, [], ins, clob)


def p1_variants():
    v = []
    for Q in range(4):
        for par in range(2):
            for sl in range(3):
                v.append((Q, par, 1, 1, 1, 2, sl))  # plain step (K request embedded, literal scalars; V_j in ring slot sl); also
                                                    # phase 1 of the step in front of the wave's diagonal tile
                v.append((Q, par, 0, 2, 1, 2, sl))  # the wave's last (diagonal) tile: masked softmax of S_j[B], no S_{j+1}
            for qk in (0, 1):
                for sm in (1, 2) if par else (1, 2, 3, 4):  # (3 / 4: tile 0 -- step 0 of a part, PAR 0)
                    v.append((Q, par, qk, sm, 1, 0, 0))     # generic steps: ring slot in the address register, requests apart
                if not PRE_ON:
                    v.append((Q, par, qk, 6, 1, 0, 0))      # ... of a sliding-window part: a tile that crosses the window's left edge (windows of at
                                                            # least two key tiles: never the diagonal as well -- code 5, both bounds, is not generated)
        v.append((Q, 1, 1, 0, 0, 0, 0))             # bare QK^T of tile 0 ("step -1": part prologue, exact-maximum pass)
        v.append((Q, 0, 1, 3, 1, 3 if Q == 0 else 2, 0))   # step 0 of a part in the embedded-request form (stream position 0, parity 0)
    return sorted(set(v))


def p2_variants():
    v = []
    for Q in range(4):
        for par in range(2):
            for sl in range(3):
                v.append((Q, par, 1, 1, 1, 2, sl))  # plain step (V request embedded, literal scalars; K_{j+2} in ring slot sl)
                v.append((Q, par, 1, 2, 1, 2, sl))  # the step in front of the wave's diagonal tile: S_{j+1}[A] masked
                v.append((Q, par, 1, 0, 0, 2, sl))  # the diagonal tile itself: O^T += V^T P^T and the V request, nothing else
                v.append((Q, par, 1, 0, 2, 2, sl))  # ... and the next part's Q rows (KR 2)
            for sm in (0, 1, 2) + (() if PRE_ON else (6,)):
                v.append((Q, par, 1, sm, 1, 0, 0))  # generic steps (6: sliding window, the left edge)
        v += [(Q, 0, 2, 0, 1, 0, 0), (Q, 0, 2, 1, 1, 0, 0), (Q, 0, 2, 2, 1, 0, 0)]   # first step of a part (O starts at 0)
        v += [(Q, 1, 0, 1, 1, 0, 0), (Q, 1, 0, 2, 1, 0, 0)]                          # part prologue: P_0[A] next to the reads of K_1
        if not PRE_ON:
            v += [(Q, 0, 2, 6, 1, 0, 0), (Q, 1, 0, 6, 1, 0, 0)]                      # sliding window: first step / prologue (of the part, or of a wave that starts late)
        v.append((Q, 0, 2, 1, 1, 2, 2))                                              # step 0 of a part, embedded-request form
    return sorted(set(v))


def gen_struct(c):
    name = f"W4Asm<{'Bf16Traits' if c.dt == 'bf16' else 'F16Traits'}, {c.D}>"
    s = f"template <> struct {name} {{\n"
    s += f"    static constexpr int KB0 = {c.KB0}, QB0 = {c.QB0}, NP = {c.NP}, NV = {c.NV};   // NV: hipcc's VGPR budget (amdgpu_num_vgpr)\n"
    s += f"    static constexpr bool PRE = {'true' if c.pre else 'false'};   // Q pre-multiplied by c, - m_ref through the MFMAs' C operand\n"
    s += f"    static constexpr int NS = {NS};   // hipcc's SGPR budget (amdgpu_num_sgpr): the streams own s[NS:NS+13]\n"
    # ---- the literal scalar registers of the embedded-request steps
    lines = [f"s_mov_b32 s{SG_KSRD}, %[klo]", f"s_and_b32 s{SG_KSRD + 1}, %[khi], 0xffff", f"s_mov_b32 s{SG_KSRD + 2}, %[nrec]", f"s_mov_b32 s{SG_KSRD + 3}, 0x00020000",
             f"s_mov_b32 s{SG_VSRD}, %[vlo]", f"s_and_b32 s{SG_VSRD + 1}, %[vhi], 0xffff", f"s_mov_b32 s{SG_VSRD + 2}, %[nrec]", f"s_mov_b32 s{SG_VSRD + 3}, 0x00020000",
             f"s_mov_b32 s{SG_KSO}, %[kso]", f"s_mov_b32 s{SG_VSO}, %[vso]"]
    s += ("    // K / V descriptors (raw buffers of nrec bytes at the heads' base addresses) and the request cursors -> literal registers\n"
          "    static __device__ __forceinline__ void set_cursors(unsigned klo, unsigned khi, unsigned vlo, unsigned vhi, unsigned nrec, unsigned kso, unsigned vso) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n")
    s += emit_asm(lines, [], ['[klo] "s"(klo)', '[khi] "s"(khi)', '[vlo] "s"(vlo)', '[vhi] "s"(vhi)', '[nrec] "s"(nrec)', '[kso] "s"(kso)', '[vso] "s"(vso)'],
                  ["scc"] + [f"s{SG_KSRD + i}" for i in range(10)], indent="        ")
    s += "#endif\n    }\n"
    s += ("    static __device__ __forceinline__ void set_offsets(unsigned kso, unsigned vso) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          f"        asm volatile(\"s_mov_b32 s{SG_KSO}, %0\\n\\ts_mov_b32 s{SG_VSO}, %1\" :: \"s\"(kso), \"s\"(vso) : \"s{SG_KSO}\", \"s{SG_VSO}\");\n#endif\n    }}\n")
    s += ("    static __device__ __forceinline__ void set_lds_base(unsigned a) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          f"        asm volatile(\"s_mov_b32 s{SG_LDSB}, %0\" :: \"s\"(a) : \"s{SG_LDSB}\");\n#endif\n    }}\n")
    # ---- phase 1
    s += ("    // SL: ring slot of the tile the statement reads, as an immediate (plain steps); 0 where the address register carries it\n"
          "    template <int Q, int PAR, int QK, int SM, int VR, int DMA, int SL = 0>\n"
          "    static __device__ __forceinline__ void p1(float c, unsigned va, int thr, unsigned dlds, __amdgpu_buffer_rsrc_t dsrd, unsigned dso,\n"
          "                                              unsigned dvo, int lo = 0) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)c; (void)va; (void)thr; (void)dlds; (void)dsrd; (void)dso; (void)dvo; (void)lo;\n"
          "        if constexpr (DMA != 0) {   // (readfirstlane: hipcc sometimes moves uniform arithmetic to the vector unit; the request wants scalars)\n"
          "            dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n            dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n        }\n")
    first = True
    for (Q, par, qk, sm, vr, dma, sl) in p1_variants():
        s += f"        {'if' if first else 'else if'} constexpr (Q == {Q} && PAR == {par} && QK == {qk} && SM == {sm} && VR == {vr} && DMA == {dma} && SL == {sl}) {{\n"
        s += gen_p1(c, Q, par, qk, sm, vr, dma, sl)
        s += "        }\n"
        first = False
    s += "        else static_assert(Q < 0, \"fa_fwd_w4_asm.inc: phase-1 variant not generated\");\n"
    s += "#endif\n    }\n"
    # ---- phase 2
    s += ("    template <int Q, int PAR, int PV, int SM, int KR, int DMA, int SL = 0>\n"
          "    static __device__ __forceinline__ void p2(float c, unsigned ka0, unsigned ka1, int thr, unsigned dlds, __amdgpu_buffer_rsrc_t dsrd,\n"
          "                                              unsigned dso, unsigned dvo, int lo = 0) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)c; (void)ka0; (void)ka1; (void)thr; (void)dlds; (void)dsrd; (void)dso; (void)dvo; (void)lo;\n"
          "        if constexpr (DMA != 0) {\n"
          "            dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n            dso = (unsigned)__builtin_amdgcn_readfirstlane((int)dso);\n        }\n")
    first = True
    for (Q, par, pv, sm, kr, dma, sl) in p2_variants():
        s += f"        {'if' if first else 'else if'} constexpr (Q == {Q} && PAR == {par} && PV == {pv} && SM == {sm} && KR == {kr} && DMA == {dma} && SL == {sl}) {{\n"
        s += gen_p2(c, Q, par, pv, sm, kr, dma, sl)
        s += "        }\n"
        first = False
    s += "        else static_assert(Q < 0, \"fa_fwd_w4_asm.inc: phase-2 variant not generated\");\n"
    s += "#endif\n    }\n"
    # ---- the wave's last tile, overlapped form: six statements (gen_diag)
    s += ("    template <int I, int PAR, int SL, int QL>\n"
          "    static __device__ __forceinline__ void diag(float c, unsigned va, int thr, unsigned dvo, unsigned ka0, unsigned ka1, __amdgpu_buffer_rsrc_t dsrd) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n        (void)c; (void)va; (void)thr; (void)dvo; (void)ka0; (void)ka1; (void)dsrd;\n")
    first = True
    for I in range(6):
        for par in range(2):
            for sl in range(3):
                for ql in ((0, 1) if I >= 4 else (0,)):
                    cond = f"I == {I} && PAR == {par} && SL == {sl}" + (f" && QL == {ql}" if I >= 4 else "")
                    s += f"        {'if' if first else 'else if'} constexpr ({cond}) {{\n" + gen_diag(c, I, par, sl, ql) + "        }\n"
                    first = False
    s += "        else static_assert(I < 0, \"fa_fwd_w4_asm.inc: diag variant not generated\");\n#endif\n    }\n"
    # ---- a stream position without arithmetic for this wave (a tile it does not see, a padding position of the part) in the
    #      embedded-request form: the tile barrier and the position's two requests with literal scalar operands, one statement.
    #      SL = position mod 3; W: the counted wait in front of the barrier (-1: none -- a padding position has no readers)
    s += "    template <int SL, int W>\n    static __device__ __forceinline__ void pad(unsigned kvo, unsigned vvo) {\n#if defined(__HIP_DEVICE_COMPILE__)\n        (void)kvo; (void)vvo;\n"
    first = True
    for sl in range(3):
        for wv in (-1, 2 * c.NP, 2 * c.NP + 2 * c.KS):
            lines = ["s_waitcnt lgkmcnt(0)" if wv < 0 else f"s_waitcnt vmcnt({wv}) lgkmcnt(0)", "s_barrier"]
            for which, slot, vo in (("k", (sl + 1) % 3, "%[kvo]"), ("v", (sl + 2) % 3, "%[vvo]")):
                for pi in range(c.NP):
                    head, load = lit_piece(c, which, slot, pi)
                    lines.append(head)
                    if pi == 0:
                        lines.append("s_nop 0")            # (an instruction between the M0 write and the request)
                    lines.append(load.replace("%[vo]", vo))
            lines += [f"s_add_u32 s{SG_KSO}, s{SG_KSO}, {c.KT}", f"s_add_u32 s{SG_VSO}, s{SG_VSO}, {c.VT}"]
            s += f"        {'if' if first else 'else if'} constexpr (SL == {sl} && W == {wv}) {{\n"
            s += emit_asm(lines, [], ['[kvo] "v"(kvo)', '[vvo] "v"(vvo)'], ["memory", "m0", "scc", f"s{SG_T}", f"s{SG_KSO}", f"s{SG_VSO}"])
            s += "        }\n"
            first = False
    s += "        else static_assert(SL < 0, \"fa_fwd_w4_asm.inc: pad variant not generated\");\n#endif\n    }\n"
    # ---- the seam: bare QK^T of the next part's tile 0 with the finished part's pack in its gaps
    s += ("    template <int Q>\n    static __device__ __forceinline__ void seam(float inv, unsigned dst) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          "        (void)inv; (void)dst;\n")
    for Q in range(4):
        s += f"        {'if' if Q == 0 else 'else if'} constexpr (Q == {Q}) {{\n" + gen_seam(c, Q) + "        }\n"
    s += "#endif\n    }\n"
    # ---- the seam: one 32-row block of O from the wave's slab to global memory, whole rows: all reads first (the pack's temporaries are
    #      free again), ONE wait, then the stores (hipcc's form -- two reads, a wait, two stores, four times over -- was ~900 cycles)
    CPR, RBP = c.RB // 16, c.RB + 16
    nb = CPR // 2
    lines = [f"ds_read_b128 v[{c.PA + 4 * i}:{c.PA + 4 * i + 3}], %[la] offset:{i * (64 // CPR) * RBP}" for i in range(nb)]
    lines.append("s_waitcnt lgkmcnt(0)")
    lines += [f"buffer_store_dwordx4 v[{c.PA + 4 * i}:{c.PA + 4 * i + 3}], %[vo{i // 4}], %[srd], 0 offen offset:{(i % 4) * 1024}" for i in range(nb)]
    s += ("    // la: LDS address of the lane's chunk of the slab (row lane / CPR, chunk lane % CPR); vo0: byte offset of that chunk of row\n"
          "    // lane / CPR of the block in the head's O, vo1 = vo0 + 4096 (D = 128: reads 4 .. 7)\n"
          "    static __device__ __forceinline__ void slab_out(__amdgpu_buffer_rsrc_t srd, unsigned la, unsigned vo0, unsigned vo1) {\n#if defined(__HIP_DEVICE_COMPILE__)\n        (void)vo1;\n")
    s += emit_asm(lines, [], ['[srd] "s"(srd)', '[la] "v"(la)', '[vo0] "v"(vo0)'] + (['[vo1] "v"(vo1)'] if nb > 4 else []),
                  ["memory"] + vregs(c.PA, 4 * nb), indent="        ")
    s += "#endif\n    }\n"
    # ---- all K fragments of one tile (ring slot in the address registers): part prologue
    lines = []
    for ks in range(c.KS):
        for h in range(2):
            lines.append(f"ds_read_b128 {c.K(ks, h)}, %[ka{ks}] offset:{h * 32 * c.RB}")
    lines.append("s_waitcnt lgkmcnt(0)")
    ins = [f'[ka{ks}] "v"(ka[{ks}])' for ks in range(c.KS)]
    s += "    static __device__ __forceinline__ void kread_all(const unsigned* ka) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], ins, ["memory"] + aregs(c.KB0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- Q fragments: rows (q0 + 32 qb + l31), 16 bytes at 32 ks + 16 hi; vo = row offset + 16 hi of block A, block B 32 rows on
    lines = ["s_nop 4"]
    for qb in range(2):
        for ks in range(c.KS):
            lines.append(f"buffer_load_dwordx4 {c.Q(qb, ks)}, %[vo{qb}], %[srd], 0 offen offset:{32 * ks}")
    s += "    static __device__ __forceinline__ void load_q(__amdgpu_buffer_rsrc_t srd, unsigned vo0, unsigned vo1) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], ['[srd] "s"(srd)', '[vo0] "v"(vo0)', '[vo1] "v"(vo1)'], ["memory"] + aregs(c.QB0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- fused query rotation (half-split pairs; K arrives rotated): the cos / sin rows of this wave's 64 queries land in the
    #      score / weight registers, which are dead between two parts (requested at the start of the previous part's epilogue).
    #      Lane (q, hi) holds d = 16 ks + 8 hi .. + 7 of its row in fragment ks; its partner d + D/2 is the same position of fragment
    #      ks + KS/2: the rotation is lane-local.  Arithmetic and rounding of rope_gfx950.hip (fa_device.h rope_pair):
    #      y1 = fma(x1, c, -(x2 s)), y2 = fma(x1, s, x2 c), one rounding to the I/O type.
    HK = c.KS // 2
    LZ = c.XA                                   # landing zone: [(qb HK + ks) 4 + {cos lo, cos hi, sin lo, sin hi}] x 4 registers
    assert LZ + 2 * HK * 16 <= c.VB0
    lines = ["s_nop 4"]
    for qb in range(2):
        for ks in range(HK):
            b = LZ + (qb * HK + ks) * 16
            for t, srd in enumerate(("%[cs]", "%[cs]", "%[ss]", "%[ss]")):
                lines.append(f"buffer_load_dwordx4 v[{b + 4 * t}:{b + 4 * t + 3}], %[vo{qb}], {srd}, 0 offen offset:{64 * ks + 16 * (t & 1)}")
    s += ("    static __device__ __forceinline__ void rope_request(__amdgpu_buffer_rsrc_t cs, __amdgpu_buffer_rsrc_t ss, unsigned vo0, unsigned vo1) {\n"
          "#if defined(__HIP_DEVICE_COMPILE__)\n")
    s += emit_asm(lines, [], ['[cs] "s"(cs)', '[ss] "s"(ss)', '[vo0] "v"(vo0)', '[vo1] "v"(vo1)'], ["memory"] + vregs(LZ, 2 * HK * 16), indent="        ")
    s += "#endif\n    }\n"
    # four dword pairs at a time, each with its own eight temporaries (the V fragment registers: V_0 is read in step 0, behind the
    # prologue), interleaved instruction by instruction: one pair alone is a dependent chain of 18 instructions
    NWAY = 4
    assert c.VB0 + 8 * NWAY <= 256

    def pair_ops(qb, ks, r, tb):
        T0, T1, X1L, X1H, X2L, X2H, TA, TB = (tb + i for i in range(8))
        bz = LZ + (qb * HK + ks) * 16
        a1 = c.QB0 + (qb * c.KS + ks) * 4 + r
        a2 = c.QB0 + (qb * c.KS + ks + HK) * 4 + r
        e = 2 * r                                   # elements e, e + 1 of the lane's eight
        cl, ch = bz + (e >> 2) * 4 + (e & 3), bz + (e >> 2) * 4 + (e & 3) + 1
        sl, sh = cl + 8, ch + 8
        o = [f"v_accvgpr_read_b32 v{T0}, a{a1}", f"v_accvgpr_read_b32 v{T1}, a{a2}"]
        if c.dt == "bf16":
            o += [f"v_lshlrev_b32 v{X1L}, 16, v{T0}", f"v_and_b32 v{X1H}, 0xffff0000, v{T0}",
                  f"v_lshlrev_b32 v{X2L}, 16, v{T1}", f"v_and_b32 v{X2H}, 0xffff0000, v{T1}"]
        else:
            o += [f"v_cvt_f32_f16 v{X1L}, v{T0}", f"v_lshrrev_b32 v{T0}, 16, v{T0}", f"v_cvt_f32_f16 v{X2L}, v{T1}",
                  f"v_lshrrev_b32 v{T1}, 16, v{T1}", f"v_cvt_f32_f16 v{X1H}, v{T0}", f"v_cvt_f32_f16 v{X2H}, v{T1}"]
        o += [f"v_mul_f32 v{TA}, v{X2L}, v{sl}", f"v_mul_f32 v{TB}, v{X2H}, v{sh}",
              f"v_mul_f32 v{X2L}, v{X2L}, v{cl}", f"v_mul_f32 v{X2H}, v{X2H}, v{ch}",
              f"v_fma_f32 v{TA}, v{X1L}, v{cl}, -v{TA}", f"v_fma_f32 v{TB}, v{X1H}, v{ch}, -v{TB}",
              f"v_fma_f32 v{X2L}, v{X1L}, v{sl}, v{X2L}", f"v_fma_f32 v{X2H}, v{X1H}, v{sh}, v{X2H}",
              f"{c.cvt} v{T0}, v{TA}, v{TB}", f"{c.cvt} v{T1}, v{X2L}, v{X2H}",
              f"v_accvgpr_write_b32 a{a1}, v{T0}", f"v_accvgpr_write_b32 a{a2}, v{T1}"]
        return o

    pairs = [(qb, ks, r) for qb in range(2) for ks in range(HK) for r in range(4)]
    lines = []
    for g0 in range(0, len(pairs), NWAY):
        lists = [pair_ops(*pairs[g0 + w], c.VB0 + 8 * w) for w in range(min(NWAY, len(pairs) - g0))]
        for i in range(max(len(l) for l in lists)):
            for l in lists:
                if i < len(l):
                    lines.append(l[i])
    lines += ["s_nop 7"]                                          # v_accvgpr_write -> MFMA operand
    s += "    static __device__ __forceinline__ void rope_rotate() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], ["memory"] + vregs(c.VB0, 8 * NWAY) + aregs(c.QB0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- negative scales (round 6): the Q fragments with their sign bits flipped -- (-q) k = -(q k) exactly, so c s with c < 0 is |c| times the scores of
    #      the negated Q, and the stream's maximum of c s (the reference's rule) is the maximum it computes anyway.  Four registers in flight.
    lines = []
    for r0 in range(0, 8 * c.KS, 4):
        lines += [f"v_accvgpr_read_b32 v{c.X + i}, a{c.QB0 + r0 + i}" for i in range(4)]
        lines += [f"v_xor_b32 v{c.X + i}, 0x80008000, v{c.X + i}" for i in range(4)]
        lines += [f"v_accvgpr_write_b32 a{c.QB0 + r0 + i}, v{c.X + i}" for i in range(4)]
    lines += ["s_nop 7"]                                          # v_accvgpr_write -> MFMA operand
    s += "    static __device__ __forceinline__ void negate_q() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += emit_asm(lines, [], [], ["memory"] + vregs(c.X, 4) + aregs(c.QB0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- LDS-DMA of this wave's pieces of one tile (everywhere but the plain step)
    lines = ["s_nop 4"]
    for i in range(c.NP):
        lines.append(f"s_add_u32 m0, %[lds], {i * 4096}")
        lines.append(f"s_add_u32 %[t], %[soff], {i * 4096}")
        lines.append("buffer_load_dwordx4 %[vo], %[srd], %[t] offen lds")
    s += "    static __device__ __forceinline__ void dma_tile(unsigned dlds, __amdgpu_buffer_rsrc_t dsrd, unsigned dsoff, unsigned dvo) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    s += "        unsigned t;\n"
    s += "        // (readfirstlane: hipcc sometimes moves uniform arithmetic to the vector unit; the request wants scalar registers)\n"
    s += "        dlds = (unsigned)__builtin_amdgcn_readfirstlane((int)dlds);\n        dsoff = (unsigned)__builtin_amdgcn_readfirstlane((int)dsoff);\n"
    s += emit_asm(lines, ['[t] "=&s"(t)'], ['[lds] "s"(dlds)', '[srd] "s"(dsrd)', '[soff] "s"(dsoff)', '[vo] "v"(dvo)'], ["memory", "m0", "scc"], indent="        ")
    s += "#endif\n    }\n"
    # ---- row maximum of tile 0: BLK 0 = S[A] (xa0, xa1), 1 = S[B] written by the bare QK^T (yb0, yb1[0]); lane-local 32 values
    s += ("    // MASKED 2 (sliding window, the exact-maximum pass): lo <= key <= thr\n"
          "    template <int BLK, int MASKED>\n    static __device__ __forceinline__ float rowmax(int thr, int lo = 0) {\n        float mx = 0.f;\n#if defined(__HIP_DEVICE_COMPILE__)\n        (void)thr; (void)lo;\n")
    first = True
    for blk in range(2):
        for masked in range(3):
            b = [c.sA(0), c.sA(1)] if blk == 0 else [c.sB(0, 0), c.sB(1, 0)]
            lines = []
            regs = []
            for h in range(2):
                for r in range(16):
                    regs.append((b[h] + r, kk(h, r)))
            if masked:
                # masked values go through two temporaries
                lines.append(f"v_mov_b32 %[mx], v{c.NINF}")
                for i in range(0, 32, 2):
                    (ra, ka_), (rb, kb_) = regs[i], regs[i + 1]
                    lines.append(f"v_cmp_le_i32 vcc, {ka_}, %[thr]")
                    lines.append(f"v_cndmask_b32 v{c.X}, v{c.NINF}, v{ra}, vcc")
                    lines.append(f"v_cmp_le_i32 vcc, {kb_}, %[thr]")
                    lines.append(f"v_cndmask_b32 v{c.X + 1}, v{c.NINF}, v{rb}, vcc")
                    if masked == 2:
                        lines.append(f"v_cmp_ge_i32 vcc, {ka_}, %[lo]")
                        lines.append(f"v_cndmask_b32 v{c.X}, v{c.NINF}, v{c.X}, vcc")
                        lines.append(f"v_cmp_ge_i32 vcc, {kb_}, %[lo]")
                        lines.append(f"v_cndmask_b32 v{c.X + 1}, v{c.NINF}, v{c.X + 1}, vcc")
                    lines.append(f"v_max3_f32 %[mx], %[mx], v{c.X}, v{c.X + 1}")
            else:
                lines.append(f"v_max3_f32 %[mx], v{regs[0][0]}, v{regs[1][0]}, v{regs[2][0]}")
                for i in range(3, 31, 2):
                    lines.append(f"v_max3_f32 %[mx], %[mx], v{regs[i][0]}, v{regs[i + 1][0]}")
                lines.append(f"v_max_f32 %[mx], %[mx], v{regs[31][0]}")
            s += f"        {'if' if first else 'else if'} constexpr (BLK == {blk} && MASKED == {masked}) {{\n"
            s += emit_asm(lines, ['[mx] "=&v"(mx)'], (['[thr] "v"(thr)'] + (['[lo] "v"(lo)'] if masked == 2 else [])) if masked else [], ["memory"] + (["vcc"] + vregs(c.X, 2) if masked else []))
            s += "        }\n"
            first = False
    s += "#endif\n        return mx;\n    }\n"
    # ---- scalars that live in literal registers
    if c.msum:    # + the ones fragment of the row-sum MFMAs: (1.0h, 1.0h) in every dword
        ones = "\\n\\t".join([f"v_mov_b32 v{c.X}, 0x3c003c00", "s_nop 0"] + [f"v_accvgpr_write_b32 a{c.ONES + i}, v{c.X}" for i in range(4)] + ["s_nop 7"])
        s += ("    static __device__ __forceinline__ void set_consts() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
              f"        asm volatile(\"v_mov_b32 v{c.NINF}, 0xff800000\\n\\t{ones}\" ::: \"v{c.NINF}\", \"v{c.X}\", "
              + ", ".join(f'\"a{c.ONES + i}\"' for i in range(4)) + ");\n#endif\n    }\n")
    else:
        s += ("    static __device__ __forceinline__ void set_consts() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
              f"        asm volatile(\"v_mov_b32 v{c.NINF}, 0xff800000\" ::: \"v{c.NINF}\");\n#endif\n    }}\n")
    s += ("    // start of a part: row sums 0\n"
          "    static __device__ __forceinline__ void zero_sums() {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          f"        asm volatile(\"v_mov_b32 v{c.L}, 0\\n\\tv_mov_b32 v{c.L + 1}, 0\\n\\tv_mov_b32 v{c.L + 2}, 0\\n\\tv_mov_b32 v{c.L + 3}, 0\" ::: "
          + ", ".join(f'"v{c.L + i}"' for i in range(4)) + ");\n#endif\n    }\n")
    def ref_asm(qb):
        lines = [f"v_mov_b32 v{c.nm(qb)}, %0"]
        cl = [f"v{c.nm(qb)}"]
        if c.pre:
            lines += [f"v_mov_b32 v{c.NMT + 16 * qb + i}, %0" for i in range(16)]
            lines += ["s_nop 1"]          # VALU write -> MFMA C operand
            cl += vregs(c.NMT + 16 * qb, 16)
        return 'asm volatile("' + "\\n\\t".join(lines) + '" :: "v"(nm) : ' + ", ".join(f'"{x}"' for x in cl) + ");"
    s += ("    // - reference of block QB (0 = A, 1 = B)\n"
          "    template <int QB>\n    static __device__ __forceinline__ void set_ref(float nm) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
          f"        if constexpr (QB == 0) {ref_asm(0)}\n"
          f"        else {ref_asm(1)}\n#endif\n    }}\n")
    # ---- pre form: Q fragments times c, rounded back (the loads have landed: the caller waited)
    lines = []
    if c.pre:
        t0, t1, t2 = f"v{c.X}", f"v{c.X + 1}", f"v{c.X + 2}"
        for r in range(8 * c.KS):
            a = f"a{c.QB0 + r}"
            lines.append(f"v_accvgpr_read_b32 {t0}, {a}")
            if c.dt == "bf16":
                lines += ["s_nop 0", f"v_and_b32 {t1}, 0xffff0000, {t0}", f"v_lshlrev_b32 {t0}, 16, {t0}"]
            else:
                lines += ["s_nop 0", f"v_lshrrev_b32 {t1}, 16, {t0}", f"v_cvt_f32_f16 {t0}, {t0}", f"v_cvt_f32_f16 {t1}, {t1}"]
            lines += [f"v_mul_f32 {t0}, %[c], {t0}", f"v_mul_f32 {t1}, %[c], {t1}", f"{c.cvt} {t2}, {t0}, {t1}", "s_nop 0", f"v_accvgpr_write_b32 {a}, {t2}"]
        lines += ["s_nop 7"]             # v_accvgpr_write -> MFMA operand
    s += "    static __device__ __forceinline__ void prescale_q(float c) {\n#if defined(__HIP_DEVICE_COMPILE__)\n        (void)c;\n"
    if c.pre:
        s += emit_asm(lines, [], ['[c] "s"(c)'], ["memory"] + vregs(c.X, 3) + aregs(c.QB0, 8 * c.KS), indent="        ")
    s += "#endif\n    }\n"
    # ---- a part that covers only a RANGE of its block's key tiles (small grids: fa_fwd_split.h) leaves its un-normalised O^T as fp32
    #      partial rows, straight from the accumulator file: lane (q, hi) owns columns 32 d + 8 g + 4 hi .. + 3 of row q in registers
    #      4 g .. 4 g + 3 of block (QB, d); vo = byte offset of the lane's row + 16 hi.  (m_ref and l: the kernel, two dwords behind the row.)
    s += "    template <int QB>\n    static __device__ __forceinline__ void partial_store(__amdgpu_buffer_rsrc_t srd, unsigned vo) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
    for qb in range(2):
        lines = ["s_nop 4"]
        for d in range(c.DB):
            for g4 in range(4):
                b = (qb * c.DB + d) * 16 + 4 * g4
                lines.append(f"buffer_store_dwordx4 a[{b}:{b + 3}], %[vo], %[srd], 0 offen offset:{(32 * d + 8 * g4) * 4}")
        s += (f"        if constexpr (QB == {qb}) {{\n" if qb == 0 else "        else {\n")
        s += emit_asm(lines, [], ['[srd] "s"(srd)', '[vo] "v"(vo)'], ["memory"])
        s += "        }\n"
    s += "#endif\n    }\n"
    if c.msum:
        # the MFMA's row of ones P^T holds the WHOLE sum of the lane's query row in both lane halves; the callers add the partner
        # half's value (the VALU form keeps one partial sum per half), so hand back one half of it: exact.  (MFMA result -> read: 19 wait states.)
        def gs(qb):
            return (f'asm volatile("s_nop 7\\n\\ts_nop 7\\n\\ts_nop 2\\n\\tv_accvgpr_read_b32 %0, a{c.LA + 16 * qb}\\n\\tv_mov_b32 %1, v{c.nm(qb)}\\n\\t'
                    f'v_mul_f32 %0, 0.5, %0" : "=&v"(l), "=v"(nm));')
        s += ("    // end of a part: row sum (from the matrix pipe, halved: see the generator) and - reference of block QB\n"
              "    template <int QB>\n    static __device__ __forceinline__ void get_sums(float& l, float& nm) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
              f"        if constexpr (QB == 0) {gs(0)}\n        else {gs(1)}\n#endif\n    }}\n")
    else:
        s += ("    // end of a part: row sum (both chains) and - reference of block QB\n"
              "    template <int QB>\n    static __device__ __forceinline__ void get_sums(float& l, float& nm) {\n#if defined(__HIP_DEVICE_COMPILE__)\n"
              f"        if constexpr (QB == 0) asm volatile(\"v_add_f32 %0, v{c.l(0, 0)}, v{c.l(0, 1)}\\n\\tv_mov_b32 %1, v{c.nm(0)}\" : \"=&v\"(l), \"=v\"(nm));\n"
              f"        else asm volatile(\"v_add_f32 %0, v{c.l(1, 0)}, v{c.l(1, 1)}\\n\\tv_mov_b32 %1, v{c.nm(1)}\" : \"=&v\"(l), \"=v\"(nm));\n#endif\n    }}\n")
    s += "};\n\n"
    return s


def main():
    hdr = ("// fa_fwd_w4_asm.inc -- GENERATED by tools/gen_w4.py (do not edit; edit the generator and re-run it).\n"
           "// Hand-placed instruction streams of the 4-wave x 64-row forward: see the generator's docstring for the register map,\n"
           "// the statement anatomy and the hazards the strings handle.  Included by fa_fwd_w4_gfx950.hip inside namespace aule_hip::{anonymous}.\n\n"
           "template <class T, int D> struct W4Asm;\n\n")
    body = ""
    for D in (128, 64):
        for dt in ("bf16", "f16"):
            body += gen_struct(Cfg(D, dt))
    with open(OUT, "w") as f:
        f.write(hdr + body)
    c = Cfg(128, "bf16")
    print(f"wrote {OUT}: {len((hdr + body).splitlines())} lines; D128 map: NV={c.NV} X={c.X} L={c.L} NM={c.NM} NINF={c.NINF} XA={c.XA} YB0={c.YB0} "
          f"YB1={c.YB1} PA={c.PA} PB={c.PB} VB0={c.VB0}")


if __name__ == "__main__":
    main()
