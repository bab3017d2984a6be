"""The D = 64 LDS image of the one-wave-per-SIMD backward kernels (fa_bwd_dkv4_gfx950.hip, fa_bwd_dq4_gfx950.hip; layout:
tools/gen_bw4.py, Cfg / chunk64), modelled on the CPU: the LDS-DMA placement (lane l of piece p writes chunk l), the row-major
ds_read_b128 addresses and the ds_read_b64_tr_b16 addresses -- lane constants as the kernels compute them, immediates as the
generators emit them -- name the same elements, and every 16-lane (b128) / 32-lane (b64) pass touches each of the 64 banks once.
No GPU: this is the address arithmetic only (the arithmetic itself is checked by tests/test_gpu_bwd.py against the fp64 judge)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

RB = 128          # bytes per row at D = 64


def _chunk(rgl, b, rr, h):       # the kernels' lambda `chunk` (fa_bwd_dkv4_gfx950.hip / fa_bwd_dq4_gfx950.hip, D = 64 branch)
    return 16 * rgl + 8 * (rgl ^ b) + 2 * rr + (h ^ b)


def _image(pbase):
    """LDS byte -> source byte (row * 128 + column byte) after the four waves' pieces of one image have landed"""
    lds = {}
    for w in range(4):
        for l in range(64):
            cd, g5 = l >> 5, l & 31
            rgl = g5 >> 4
            cb = rgl ^ ((g5 >> 3) & 1)
            rr, ch = (g5 >> 1) & 3, (g5 & 1) ^ cb
            src = (w * 8 + 4 * rgl + rr) * RB + (2 * cd + cb) * 32 + ch * 16      # vost[0] of wave w, lane l
            for i in range(16):
                lds[pbase[w] + 16 * l + i] = src + i                              # M0 = slot + 1040 w, lane-linear
    return lds


def _offsets(lines):
    return [int(re.search(r"offset:(\d+)", ln).group(1)) for ln in lines]


def test_chunk64_is_the_kernels_chunk():
    import gen_bw4
    for rgl in range(2):
        for d in range(2):
            for b in range(2):
                for rr in range(4):
                    for h in range(2):
                        assert gen_bw4.chunk64(rgl, d, b, rr, h) == 32 * d + _chunk(rgl, b, rr, h)


def test_d64_image_addresses_and_banks():
    import gen_bw4
    import gen_dq4
    c = gen_bw4.Cfg(64, "bf16")
    cq = gen_dq4.Cfg(64, "bf16")
    assert (c.PBASE, c.IMG, c.NP) == (cq.PBASE, cq.IMG, cq.NP) == ([0, 1040, 2080, 3120], 4352, 2)
    lds = _image(c.PBASE)
    assert len(set(lds.values())) == 32 * RB                  # every byte of the 32 x 64 block exactly once
    assert max(lds) < c.IMG

    # ---- row-major fragments: lane (row l31, hi) reads columns 16 ks + 8 hi .. + 7 of its row
    #      dK/dV kernel: p2's reads of statement q (k-slice q of Q at +0 and of dO at +IMG); dQ kernel: rm_reads(ks)
    for ks in range(4):
        reg = "ra1" if ks & 1 else "ra"
        # the statement of the dK/dV stream that reads k-slice ks
        txt = gen_bw4.gen_p2(c, ks, 0, 0, 1, 0, 0)
        offs = sorted(set(_offsets([ln for ln in txt.split("\n") if "ds_read_b128" in ln])))
        assert all(f"%[{reg}]" in ln for ln in txt.split("\n") if "ds_read_b128" in ln)
        kq, vq = gen_dq4.rm_reads(cq, ks, "%[ra]")
        assert ("%[rab]" in kq) == bool(ks & 1)
        assert offs == sorted(_offsets([kq, vq])) == [512 * (ks >> 1), c.IMG + 512 * (ks >> 1)]
        imm = 512 * (ks >> 1)
        for p0 in range(4):                                   # the four 16-lane passes of a ds_read_b128
            banks = []
            for lane in range(16 * p0, 16 * p0 + 16):
                l31, hi = lane & 31, lane >> 5
                a = 1040 * (l31 >> 3) + 16 * _chunk((l31 >> 2) & 1, ks & 1, l31 & 3, hi) + imm      # a_sub / a_sub1 + immediate
                for i in range(16):
                    assert lds[a + i] == l31 * RB + ks * 32 + hi * 16 + i, (lane, ks)
                banks += [(a // 4 + j) % 64 for j in range(4)]
            assert sorted(banks) == list(range(64)), (ks, p0)

    # ---- transpose reads: lane L reads 8 bytes: row 16 kk + 8 e + 4 hi + ((L >> 2) & 3), columns 32 d + 16 ((L >> 4) & 1) + 4 (L & 3) ..
    for kk in range(2):
        for d in range(2):
            st = c.DB * kk + d
            lines = gen_bw4.tr_reads(c, st)
            o_g0, o_g1, o_q0, o_q1 = _offsets(lines)              # dO e = 0, 1 (+ IMG), Q e = 0, 1
            assert (o_g0 - c.IMG, o_g1 - c.IMG) == (o_q0, o_q1)
            assert _offsets(gen_dq4.tr_reads(cq, kk, d)) == [o_q0, o_q1]
            for e, imm in ((0, o_q0), (1, o_q1)):
                assert imm == c.PBASE[2 * kk + e] + 512 * d
                for p0 in range(2):                           # the two 32-lane passes of a ds_read_b64
                    banks = []
                    for lane in range(32 * p0, 32 * p0 + 32):
                        hi = lane >> 5
                        a = 16 * _chunk(hi, (lane >> 4) & 1, (lane >> 2) & 3, (lane >> 1) & 1) + (lane & 1) * 8 + imm      # tr_off + immediate
                        row = 16 * kk + 8 * e + 4 * hi + ((lane >> 2) & 3)
                        col = 32 * d + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)
                        for i in range(8):
                            assert lds[a + i] == row * RB + col * 2 + i, (lane, kk, e, d)
                        banks += [(a // 4 + j) % 64 for j in range(2)]
                    assert sorted(banks) == list(range(64)), (kk, e, d, p0)
