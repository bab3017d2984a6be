"""RoPE (second half) on the GPU: the rotation pass csrc/rope_gfx950.hip behind
aule.flash_attention_rope / apply_rope_separate (half-split pairs, python/aule/triton_flash.py:561-703) and behind
the rot_cos / rot_sin handles of aule_attention_forward_gpu (interleaved pairs, shaders/attention_f32.comp:98-111).

  * golden vectors recorded from the reference (tests/golden/rope_*.npz: its tables, its separate rotation, its
    FA-2 kernel over the rotated Q, K -- what its own self-test holds the fused path to);
  * the pass alone against the fp64 oracle: both layouts, inverse, position offset, every dtype, vector and scalar
    paths (head_dim 6 / 20 / 80), in place;
  * RoPE + attention forward AND backward against the oracle chain rotate -> attention -> rotate back (the
    reference's backward ignores the rotation, so there is no reference golden for the gradients);
  * properties: rotate then rotate back is the identity; zero angles are plain attention bit for bit; scores depend
    on relative position only (shifting every position by the same offset leaves the output unchanged).
"""
import math
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden
from util import BWD_TOL, LSE_TOL, assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype="fp32"):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))


@pytest.mark.parametrize("path", golden_files("rope_"), ids=lambda p: p.split("/")[-1][:-4])
def test_rope_goldens(path):
    import torch
    import aule
    g = load_golden(path)
    raw = np.load(path)
    cos, sin = _dev(torch, raw["cos"]), _dev(torch, raw["sin"])
    q, k, v = (_dev(torch, g[x]) for x in ("q", "k", "v"))
    out = aule.flash_attention_rope(q, k, v, cos, sin, causal=g["causal"])
    assert_close(out.cpu().numpy(), g["out"], 2e-5, 1e-5, g["name"] + " out")
    out3 = aule.flash_attention_rope(q, k, v, cos[None], sin[None], causal=g["causal"])     # [1, S, D/2] tables
    assert torch.equal(out, out3)
    qr, kr = aule.apply_rope_separate(q, k, cos, sin)
    assert_close(qr.cpu().numpy(), raw["q_rot"], 1e-5, 1e-5, g["name"] + " q_rot")
    assert_close(kr.cpu().numpy(), raw["k_rot"], 1e-5, 1e-5, g["name"] + " k_rot")
    D = q.shape[-1]
    c2, s2 = aule.precompute_rope_frequencies(raw["cos"].shape[0], D, device="cuda")
    assert_close(c2.cpu().numpy(), raw["cos"], 1e-5, 1e-5, "cos table")
    assert_close(s2.cpu().numpy(), raw["sin"], 1e-5, 1e-5, "sin table")


PASS_CASES = [  # dtype, B, H, S, D, layout, inverse, pos_offset
    ("fp32", 2, 3, 50, 64, "half", False, 0),
    ("fp32", 1, 2, 33, 128, "interleaved", False, 5),
    ("bf16", 2, 4, 100, 128, "half", False, 0),
    ("bf16", 1, 2, 64, 64, "interleaved", True, 3),
    ("fp16", 1, 3, 77, 32, "half", True, 11),
    ("fp16", 1, 1, 40, 80, "half", False, 0),         # D/2 = 40: 8-wide vector path
    ("bf16", 1, 2, 19, 20, "half", False, 2),         # D/2 = 10: scalar path
    ("fp32", 1, 2, 19, 6, "interleaved", False, 0),   # scalar path
    ("fp32", 1, 1, 9, 20, "half", True, 0),
]


@pytest.mark.parametrize("case", PASS_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_rope_pass_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, H, S, D, layout, inverse, off = case
    rng = np.random.RandomState(31)
    x = quantize(rng.randn(B, H, S, D).astype(np.float32), dtype)
    cos, sin = oracle_mod.rope_tables(S + off + 4, D)
    want = oracle_mod.rope_f64(x, cos, sin, layout, inverse, off)
    tx, tc, ts = _dev(torch, x, dtype), _dev(torch, cos), _dev(torch, sin)
    got = at.rope_raw(tx, tc, ts, layout, inverse, off)
    tol = {"fp32": 2e-6, "fp16": 2e-3, "bf16": 1.6e-2}[dtype]      # one rounding to the I/O dtype
    assert_close(got.float().cpu().numpy(), want, tol * max(1.0, float(np.abs(want).max())), 0, "rope")
    assert torch.equal(tx, _dev(torch, x, dtype))                   # out of place left the input alone
    at.rope_raw(tx, tc, ts, layout, inverse, off, out=tx)            # in place
    assert torch.equal(tx, got)


ATTN_CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, causal, window
    ("bf16", 1, 4, 2, 300, 300, 128, True, -1),
    ("bf16", 2, 2, 2, 257, 400, 64, False, -1),
    ("fp16", 1, 4, 1, 200, 200, 32, True, 50),
    ("fp32", 1, 2, 2, 130, 130, 64, True, -1),
    ("fp32", 1, 2, 1, 70, 150, 80, False, -1),          # head_dim padded to 128 after the rotation
    ("bf16", 1, 4, 4, 100, 260, 128, "bottom-right", -1),   # queries at positions Sk - Sq + i
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_rope_attention_forward_backward_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, W = case
    rng = np.random.RandomState(17)
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), dtype)
                   for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    cos, sin = oracle_mod.rope_tables(max(Sq, Sk), D)
    qoff = Sk - Sq if causal == "bottom-right" else 0
    # the judge: rotate (rounded to the I/O dtype, as the pass stores it), attend, rotate the gradients back
    qr = quantize(oracle_mod.rope_f64(q, cos, sin, "half", False, qoff), dtype)
    kr = quantize(oracle_mod.rope_f64(k, cos, sin, "half"), dtype)
    ref, _ = oracle_mod.fwd_f64(qr, kr, v, causal, None, W)
    rq, rk, rv = oracle_mod.bwd_f64(qr, kr, v, do, causal, None, W)
    rq = oracle_mod.rope_f64(rq, cos, sin, "half", True, qoff)
    rk = oracle_mod.rope_f64(rk, cos, sin, "half", True)
    tq, tk, tv = (_dev(torch, x, dtype).requires_grad_(True) for x in (q, k, v))
    out = at.flash_attention_rope_hip(tq, tk, tv, _dev(torch, cos), _dev(torch, sin), causal=causal, window=W)
    out.backward(_dev(torch, do, dtype))
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    assert_close(out.detach().float().cpu().numpy(), ref, atol, rtol, "out")
    a, r = BWD_TOL[dtype]
    for name, got, want in (("dq", tq.grad, rq), ("dk", tk.grad, rk), ("dv", tv.grad, rv)):
        assert_close(got.float().cpu().numpy(), want, a * max(1.0, float(np.abs(want).max())), r, name)


def test_rope_properties():
    import torch
    import aule
    from aule import _torch as at
    torch.manual_seed(4)
    B, H, S, D = 1, 4, 256, 128
    q = torch.randn(B, H, S, D, device="cuda", dtype=torch.float32)
    k = torch.randn(B, H, S, D, device="cuda", dtype=torch.float32)
    v = torch.randn(B, H, S, D, device="cuda", dtype=torch.float32)
    cos, sin = aule.precompute_rope_frequencies(S + 64, D, device="cuda")
    for layout in ("half", "interleaved"):
        back = at.rope_raw(at.rope_raw(q, cos, sin, layout), cos, sin, layout, inverse=True)
        assert float((back - q).abs().max()) < 1e-5
        # a rotation: norms of the pairs are kept
        assert torch.allclose(at.rope_raw(q, cos, sin, layout).pow(2).sum(-1), q.pow(2).sum(-1), rtol=1e-5, atol=1e-4)
    # zero angles: plain attention, bit for bit
    one, zero = torch.ones_like(cos), torch.zeros_like(sin)
    assert torch.equal(aule.flash_attention_rope(q, k, v, one, zero, causal=True), aule.flash_attention(q, k, v, causal=True))
    # relative positions only: every position shifted by 37 -> same scores, same output
    base = aule.flash_attention_rope(q, k, v, cos, sin, causal=True)
    qs, ks = at.rope_raw(q, cos, sin, "half", pos_offset=37), at.rope_raw(k, cos, sin, "half", pos_offset=37)
    shifted = aule.flash_attention(qs, ks, v, causal=True)
    assert float((shifted - base).abs().max()) < 2e-4


def test_rope_through_the_c_abi_handles(oracle_mod):
    """tests/test_rope_unit.py of the reference: Aule().attention_gpu with rot_cos / rot_sin [1, 1, S, D/2],
    interleaved pairs, against rotate-then-attend."""
    from aule.hip import Aule
    np.random.seed(42)                                              # test_rope_unit.py:18-21
    B, H, S, D = 1, 1, 8, 64
    q, k, v = (np.random.randn(B, H, S, D).astype(np.float32) for _ in range(3))
    half = D // 2
    freqs = 1.0 / (10000 ** (np.arange(0, half, dtype=np.float32) / half))      # :24-31
    emb = np.outer(np.arange(S, dtype=np.float32), freqs)
    cos, sin = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)
    with Aule() as a:
        # Dirty the allocator first: head_dim 48 is stored with a row pitch of 64 and the kernels need the pad to be
        # zero, so the rotated copies must not inherit freed memory (this passed by luck on fresh zero pages once).
        for _ in range(4):
            junk = [a.tensor((1, 4, 96, 64)) for _ in range(6)]
            for t in junk:
                t.upload(np.full((1, 4, 96, 64), np.nan, dtype=np.float32))
            for t in junk:
                t.destroy()
                a._tensors.remove(t)
        out = a.attention(q, k, v, rot_cos=cos.reshape(1, 1, S, half), rot_sin=sin.reshape(1, 1, S, half), causal=False)
        base = a.attention(q, k, v, causal=False)
        # a GQA / cross-attention case with a longer table and a 2-D table
        rng = np.random.RandomState(6)
        q2, k2, v2 = rng.randn(1, 4, 50, 48).astype(np.float32), rng.randn(1, 2, 90, 48).astype(np.float32), rng.randn(1, 2, 90, 48).astype(np.float32)
        c2, s2 = oracle_mod.rope_tables(96, 48)
        out2 = a.attention(q2, k2, v2, rot_cos=c2, rot_sin=s2, causal=True)
        with pytest.raises(Exception):
            a.attention(q, k, v, rot_cos=cos[:4].reshape(1, 1, 4, half), rot_sin=sin[:4].reshape(1, 1, 4, half))  # short
        # "[1, 1, S, D/2] or similar broadcastable" (the reference's engine reads the table as a flat [positions, D/2] buffer):
        # the position axis may sit in any of the three leading dimensions ...
        alt = [a.attention(q, k, v, rot_cos=cos.reshape(shp), rot_sin=sin.reshape(shp), causal=False)
               for shp in ((1, S, 1, half), (S, 1, 1, half))]
        # ... but a per-head table is not one table: refused, not read across heads
        with pytest.raises(Exception):
            a.attention(q, k, v, rot_cos=np.tile(cos, (2, 1)).reshape(1, 2, S, half), rot_sin=np.tile(sin, (2, 1)).reshape(1, 2, S, half))
    qr, kr = oracle_mod.rope_f64(q, cos, sin, "interleaved"), oracle_mod.rope_f64(k, cos, sin, "interleaved")
    ref, _ = oracle_mod.fwd_f64(qr, kr, v, False)
    assert_close(out, ref, 1e-5, 1e-5, "rope handles")              # the reference's own bar is 1e-3 (:104)
    for o in alt:
        assert np.array_equal(o, out)
    ref_base, _ = oracle_mod.fwd_f64(q, k, v, False)
    assert_close(base, ref_base, 1e-5, 1e-5, "no rope")
    ref2, _ = oracle_mod.fwd_f64(oracle_mod.rope_f64(q2, c2, s2, "interleaved"), oracle_mod.rope_f64(k2, c2, s2, "interleaved"),
                                 v2, True)
    assert_close(out2, ref2, 1e-5, 1e-5, "rope handles gqa")


def test_rope_rejections():
    import torch
    import aule
    q = torch.randn(1, 2, 64, 64, device="cuda", dtype=torch.bfloat16)
    cos, sin = aule.precompute_rope_frequencies(32, 64, device="cuda")
    with pytest.raises(ValueError):
        aule.flash_attention_rope(q, q, q, cos, sin)                    # table shorter than the sequence
    c2, s2 = aule.precompute_rope_frequencies(64, 32, device="cuda")
    with pytest.raises(ValueError):
        aule.flash_attention_rope(q, q, q, c2, s2)                      # wrong head_dim // 2
    with pytest.raises(ValueError):
        aule.flash_attention_rope(q, q, q, None, None)
    q7 = torch.randn(1, 1, 8, 7, device="cuda")
    with pytest.raises(ValueError):
        aule.flash_attention_rope(q7, q7, q7, cos[:, :3], sin[:, :3])   # odd head_dim


FUSED_CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, causal
    ("bf16", 2, 8, 8, 1024, 1024, 128, True),        # two parts per workgroup list entry: prologue and seam rotations
    ("bf16", 4, 32, 32, 2048, 2048, 128, True),      # the C2-like shape: several parts per workgroup
    ("bf16", 1, 8, 2, 777, 1300, 128, False),        # ragged last block (rows >= Sq index past nothing: they read 0)
    ("fp16", 2, 4, 4, 600, 600, 64, True),           # D = 64 instances (two workgroups per CU)
    ("fp16", 16, 16, 4, 300, 2048, 128, "bottom-right"),  # queries at positions Sk - Sq + i (enough pairs of Q blocks that the
                                                          # un-fused launch stays on the plain stream too: same kernel, same order)
    ("bf16", 1, 4, 4, 512, 512, 64, False),
    # round 6: the sliding-window instances rotate Q in their part prologue like the plain ones (waves that start late included)
    ("bf16", 2, 8, 8, 2048, 2048, 128, True, 256),
    ("fp16", 1, 8, 2, 1500, 1500, 64, True, 300),
]


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_fused_query_rotation_is_the_separate_pass_bit_for_bit(case, oracle_mod, monkeypatch):
    """aule_attention_forward_rope_ex (Q rotated in the forward kernel's registers, K by the pass) against
    aule_rope_ex(Q) + aule_rope_ex(K) + aule_attention_forward_ex: the same arithmetic and rounding, so the outputs are
    EQUAL; and both against the fp64 oracle chain.  The Python inference path takes the fused route by itself."""
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, causal = case[:8]
    W = case[8] if len(case) > 8 else -1
    rng = np.random.RandomState(23)
    q, k, v = (quantize(rng.randn(*s).astype(np.float32), dtype) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D)))
    cos, sin = oracle_mod.rope_tables(max(Sq, Sk) + 3, D)
    qoff = Sk - Sq if causal == "bottom-right" else 0
    tq, tk, tv, tc, ts = _dev(torch, q, dtype), _dev(torch, k, dtype), _dev(torch, v, dtype), _dev(torch, cos), _dev(torch, sin)
    code = at.causal_code(causal)
    # the one-wave-per-SIMD kernel rotates Q itself (table rows land in its score registers between two parts): fusable == route 8 (fa_fwd_gfx950.hip)
    fus = at.rope_fusable(tq, tk, code, W, tc, ts, qoff)
    assert fus
    sc = 1.0 / math.sqrt(D)
    kr = at.rope_raw(tk, tc, ts, "half", False, 0)
    qr = at.rope_raw(tq, tc, ts, "half", False, qoff)
    two_pass, lse = at.fwd_raw(qr, kr, tv, code, sc, want_lse=True, window=W)
    if fus:
        fused, lse = at.fwd_raw(tq, kr, tv, code, sc, want_lse=True, q_rope=(tc, ts, qoff), window=W)
        assert torch.equal(fused, two_pass)
    with torch.no_grad():
        auto = at.flash_attention_rope_hip(tq, tk, tv, tc, ts, causal=causal, window=W)
    assert torch.equal(auto, two_pass)
    monkeypatch.setenv("AULE_HIP_ROPE_FUSE", "0")
    with torch.no_grad():
        assert torch.equal(at.flash_attention_rope_hip(tq, tk, tv, tc, ts, causal=causal, window=W), two_pass)
    if B * Hq * Sq * Sk <= 2 * 8 * 1024 * 1024:
        qo = quantize(oracle_mod.rope_f64(q, cos, sin, "half", False, qoff), dtype)
        ko = quantize(oracle_mod.rope_f64(k, cos, sin, "half"), dtype)
        ref, ref_lse = oracle_mod.fwd_f64(qo, ko, v, causal, None, W)
        atol, rtol = fwd_tol(dtype, np.abs(v).max())
        assert_close(two_pass.float().cpu().numpy(), ref, atol, rtol, "out")
        assert_close(lse.cpu().numpy(), ref_lse, LSE_TOL[dtype], 0, "lse")


def test_fused_query_rotation_rejections():
    """Configurations the kernel does not rotate for are refused loudly (-3), never computed un-rotated."""
    import torch
    from aule import _capi, _torch as at
    q = torch.randn(1, 4, 512, 32, device="cuda", dtype=torch.bfloat16)       # D = 32: no fused instance
    cos = torch.ones(512, 16, device="cuda"); sin = torch.zeros(512, 16, device="cuda")
    assert not at.rope_fusable(q, q, 1, -1, cos, sin, 0)
    with pytest.raises(_capi.AuleError, match="not fused"):
        at.fwd_raw(q, q, q, 1, 0.2, q_rope=(cos, sin, 0))
    q = torch.randn(1, 4, 512, 128, device="cuda", dtype=torch.bfloat16)
    cos = torch.ones(500, 64, device="cuda"); sin = torch.zeros(500, 64, device="cuda")   # table shorter than Sq
    with pytest.raises(_capi.AuleError, match="not fused"):
        at.fwd_raw(q, q, q, 1, 0.1, q_rope=(cos, sin, 0))


@pytest.mark.parametrize("dtype,B,Hq,Hkv,S,D,causal,mag", [
    ("bf16", 4, 32, 8, 2048, 128, True, 5.0),      # several parts per workgroup, some of them re-run
    ("fp16", 8, 32, 16, 768, 128, True, 6.0),      # fp16: weights overflow the fixed reference
    ("bf16", 4, 16, 16, 1024, 64, False, 10.0),    # D = 64
])
def test_fused_rotation_through_the_exact_maximum_stream(dtype, B, Hq, Hkv, S, D, causal, mag):
    """Inputs scaled until the fixed-reference range verdict fails (the one-wave-per-SIMD kernel then re-runs those parts in a
    second stream whose exact-maximum pass has to rotate Q as well): fused == rotation pass + plain forward, bit for bit, on
    full grids (a small grid would take the split route for the plain call and round differently), and finite."""
    import torch
    import aule
    from aule import _torch as at
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    g = torch.Generator(device="cuda").manual_seed(3)
    q = (torch.randn(B, Hq, S, D, device="cuda", generator=g) * mag).to(dt)
    k = (torch.randn(B, Hkv, S, D, device="cuda", generator=g) * mag).to(dt)
    v = torch.randn(B, Hkv, S, D, device="cuda", generator=g).to(dt)
    cos, sin = aule.precompute_rope_frequencies(S, D, device="cuda")
    cos, sin = cos.contiguous(), sin.contiguous()
    code = 1 if causal else 0
    assert at.rope_fusable(q, k, code, -1, cos, sin, 0)
    sc = 1.0 / math.sqrt(D)
    kr, qr = at.rope_raw(k, cos, sin), at.rope_raw(q, cos, sin)
    two, lse2 = at.fwd_raw(qr, kr, v, code, sc, want_lse=True)
    fused, lse1 = at.fwd_raw(q, kr, v, code, sc, want_lse=True, q_rope=(cos, sin, 0))
    assert torch.isfinite(fused.float()).all() and torch.isfinite(lse1).all()
    assert torch.equal(fused, two) and torch.equal(lse1, lse2)


def test_negative_scale_with_the_fused_rotation():
    """Round 5 (found by tools/fuzz_parity.py split): flash_attention_rope with scale < 0 used to pick the fused rotation although the kernel that
    rotates Q did not take negative scales (the helper did not look at the scale), and the launch refused it.  Round 6: the kernel takes negative
    scales -- the Q fragments are negated in registers after the rotation, c = |scale| log2(e) -- so the inference call is fused again, and still
    the same bits as rotating by hand and calling the plain forward."""
    import torch
    import aule
    from aule import _torch as at
    g = torch.Generator(device="cuda").manual_seed(8)
    B, Hq, Hkv, S, D = 1, 8, 2, 1500, 128
    q, k, v = (torch.randn(B, h, S, D, device="cuda", dtype=torch.bfloat16, generator=g) for h in (Hq, Hkv, Hkv))
    cos, sin = aule.precompute_rope_frequencies(S, D, device="cuda")
    cos, sin = cos.contiguous(), sin.contiguous()
    assert at.rope_fusable(q, k, 1, -1, cos, sin, 0) and at.rope_fusable(q, k, 1, -1, cos, sin, 0, -0.2)
    with torch.no_grad():
        out = aule.flash_attention_rope(q, k, v, cos, sin, causal=True, scale=-0.2)
        ref = aule.flash_attention(at.rope_raw(q, cos, sin), at.rope_raw(k, cos, sin), v, causal=True, scale=-0.2)
    assert torch.isfinite(out.float()).all() and torch.equal(out, ref)
