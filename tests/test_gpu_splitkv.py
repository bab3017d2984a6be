"""Short queries against long K/V (SURVEY.md 8d points C5b, C5c and the decode regime): non-causal 16-bit problems
that the plain tiled launch would run badly.  Two kernels serve them, chosen by a measured rule
(csrc/fa_fwd_gfx950.hip short_query_route, pinned on CPU by tests/test_capi_symbols.py::test_forward_routing_rule):

  route 5  the tiled kernel with GQA groups packed into rows and the key range split over workgroups
           (fa_fwd_pp_gfx950.hip SPLIT instances + fa_fwd_splitkv_combine) -- most shapes, including C5b / C5c;
  route 4  the wave-per-chunk split-KV kernel (fa_fwd_splitkv_gfx950.hip) -- the streaming corner: >= 32 units,
           <= 16 packed rows, K+V >= 100 MB (and every paged decode, tests/test_gpu_paged.py).

Every case states the route it is meant to exercise and asserts it, so a change of the rule cannot silently move a
kernel out of coverage.  Parity vs the fp64 oracle: MQA/GQA/MHA packing, partial row tiles, ragged Sk, every head_dim,
negative scale, Sq > 64, and the autograd round trip (the backward consumes the LSE these paths wrote)."""
import math

import numpy as np
import pytest

from util import BWD_TOL, LSE_TOL, assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu

CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, scale, route
    ("fp16", 1, 32, 1, 1, 16384, 64, None, 5),      # C5b
    ("fp16", 1, 32, 1, 64, 16384, 64, None, 5),     # C5c
    ("bf16", 2, 8, 2, 1, 4096, 128, None, 5),
    ("bf16", 1, 4, 4, 3, 1500, 128, None, 5),       # MHA: 3 rows per unit, ragged Sk
    ("fp16", 2, 6, 3, 17, 2049, 32, 0.3, 5),
    ("bf16", 1, 16, 2, 9, 1024, 64, -0.2, 5),       # 72 packed rows (last row tile partial), negative scale
    ("bf16", 3, 2, 2, 64, 5000, 128, None, 5),
    ("bf16", 1, 8, 8, 300, 9000, 128, None, 5),     # Sq > 64: two Q blocks per head, ragged last split
    ("fp16", 2, 4, 2, 129, 2000, 128, None, 5),     # 258 packed rows: second Q block holds 2 rows
    ("bf16", 16, 32, 8, 1, 4096, 128, None, 5),     # large-batch decode (was the plain tiled kernel: 4 of 256 rows used)
    ("bf16", 8, 32, 8, 1, 8192, 128, None, 4),      # the wave kernel's corner: 64 units, 4 packed rows, 268 MB of K+V
    ("fp16", 8, 32, 8, 4, 16384, 64, 0.2, 4),       # ... 16 packed rows, D = 64
]


def _route(case):
    """Kernel the dispatcher picks for `case` (aule_hip_debug_forward_route: 4 = split-KV)."""
    import ctypes
    from aule import _capi
    dtype, B, Hq, Hkv, Sq, Sk, D = case[:7]
    lib = _capi.get_lib()
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = 0, -1
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_splitkv_forward_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, scale, want_route = case
    rng = np.random.RandomState(21)
    q, k, v = (quantize(rng.randn(*s).astype(np.float32), dtype) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D)))
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a: torch.from_numpy(a).to("cuda", torch_dtype(dtype))
    assert _route(case) == want_route, "the dispatch rule moved this shape: it no longer covers the kernel it was written for"
    out, lse = at.fwd_raw(dev(q), dev(k), dev(v), False, sc)
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, False, scale)
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    assert_close(out.float().cpu().numpy(), ref, atol, rtol, "out")
    assert_close(lse.cpu().numpy(), ref_lse, LSE_TOL[dtype], 1e-5, "lse")


def test_splitkv_feeds_the_backward(oracle_mod):
    import torch
    import aule
    rng = np.random.RandomState(22)
    B, Hq, Hkv, Sq, Sk, D = 1, 8, 2, 5, 1536, 64
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), "bf16")
                   for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    tq, tk, tv = (torch.from_numpy(x).to("cuda", torch.bfloat16).requires_grad_(True) for x in (q, k, v))
    out = aule.flash_attention(tq, tk, tv, causal=False)
    out.backward(torch.from_numpy(do).to("cuda", torch.bfloat16))
    rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, False)
    a, r = BWD_TOL["bf16"]
    for name, got, want in (("dq", tq.grad, rq), ("dk", tk.grad, rk), ("dv", tv.grad, rv)):
        assert_close(got.float().cpu().numpy(), want, a * max(1.0, float(np.abs(want).max())), r, name)


# ---- small grids: route 7, the one-wave-per-SIMD forward with every pair of causal Q blocks (every non-causal block) cut into key
#      ranges (fa_fwd_w4_gfx950.hip part table + fa_fwd_combine, fa_fwd_split.h).  The plan is arithmetic (split_cuts): which blocks
#      are cut, and where.  (Rounds 2-3 ran this on the two-waves-per-SIMD stream kernel; a negative scale now takes the ping-pong
#      kernel, whole pairs.)
CAUSAL_SPLIT_CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale, spike
    ("bf16", 1, 4, 4, 2048, 2048, 128, True, None, 0),              # 8 Q blocks: 4 pairs, every far block cut
    ("bf16", 1, 8, 2, 1792, 1792, 128, True, None, 0),              # 7 Q blocks: the middle block is a pair of its own
    ("bf16", 1, 2, 2, 1900, 1900, 128, True, None, 0),              # ragged last block (rows >= Sq never stored)
    ("fp16", 2, 4, 4, 2048, 2048, 64, True, 0.2, 0),                # D = 64 instances
    ("bf16", 1, 8, 8, 1024, 4096, 128, "bottom-right", None, 0),    # queries at positions Sk - Sq + i: 7 pieces per pair
    ("fp16", 1, 4, 1, 1000, 3000, 128, "bottom-right", -0.1, 0),    # ragged Sq and Sk, negative scale, 5 pieces
    ("bf16", 1, 8, 8, 4096, 4096, 128, True, None, 0),              # 4 pieces: both blocks of the middle pairs are cut
    ("bf16", 1, 8, 8, 2048, 2048, 128, False, None, 0),             # non-causal: single blocks cut in two (no pairing)
    ("fp16", 1, 16, 16, 1100, 3333, 64, False, 0.15, 0),            # non-causal, ragged Sq / Sk, three pieces, D = 64
    ("bf16", 1, 4, 4, 2048, 2048, 128, True, None, 40.0),           # a spiked key late in the far ranges: the fixed-reference
    ("fp16", 1, 4, 4, 2048, 2048, 128, True, None, 40.0),           #   verdict fails on partial parts -> re-run, same partial slot
]


def _route_causal(dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale=0.0):
    import ctypes
    from aule import _capi, _torch as at
    lib = _capi.get_lib()
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = at.causal_code(causal), -1
    d.scale = scale
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))


@pytest.mark.parametrize("case", CAUSAL_SPLIT_CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_causal_split_forward_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, causal, scale, spike = case
    rng = np.random.RandomState(29)
    q, k, v = (rng.randn(*s).astype(np.float32) for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D)))
    if spike:
        # one key per head, in the SECOND key range of the far blocks, aligned with the last queries: their logits exceed
        # the first tile's maximum by far more than the fixed reference tolerates
        j = Sk - 300
        k[:, :, j, :] = spike * q[:, ::Hq // Hkv, Sq - 5, :] / np.linalg.norm(q[:, ::Hq // Hkv, Sq - 5, :], axis=-1, keepdims=True)
    q, k, v = (quantize(x, dtype) for x in (q, k, v))
    sc = (1 / math.sqrt(D)) if scale is None else scale
    dev = lambda a: torch.from_numpy(a).to("cuda", torch_dtype(dtype))
    want = 7   # (round 6: a negative scale takes the same route, on negated Q fragments)
    assert _route_causal(dtype, B, Hq, Hkv, Sq, Sk, D, causal, sc) == want, "the dispatch rule moved this shape off the key-range split"
    out, lse = at.fwd_raw(dev(q), dev(k), dev(v), at.causal_code(causal), sc)
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, causal, scale)
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    got = out.float().cpu().numpy()
    assert_close(got, ref, atol, rtol, "out")
    assert_close(lse.cpu().numpy(), ref_lse, LSE_TOL[dtype], 1e-5, "lse")
    print("causal split %s achieved: out max|err| %.3e, lse max|err| %.3e" % (case[:8], np.abs(got - ref).max(),
                                                                             np.abs(lse.cpu().numpy() - ref_lse).max()))


def test_causal_split_at_the_shape_it_was_built_for(oracle_mod):
    """B1 H8 S8192 D128 bf16 causal (128 paired items on 256 CUs): sampled rows vs the fp64 judge, and the backward fed by
    the LSE the merge kernel wrote."""
    import torch
    import aule
    from aule import _torch as at
    B, H, S, D = 1, 8, 8192, 128
    assert _route_causal("bf16", B, H, H, S, S, D, True) == 7
    gen = torch.Generator(device="cuda").manual_seed(99)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=gen) for _ in range(3))
    out, lse = at.fwd_raw(q, k, v, 1, 1.0 / math.sqrt(D))
    rng = np.random.RandomState(6)
    rows = np.unique(np.concatenate([rng.randint(0, B * H * S, 40), [0, 255, 256, S - 1, 4095, 4096, 4096 + 255, (H - 1) * S + 4352,
                                                                     H * S - 1, 16 * 256, 16 * 256 - 1]]))
    ref, ref_lse = oracle_mod.fwd_rows_f64(q.float().cpu().numpy(), k.float().cpu().numpy(), v.float().cpu().numpy(), rows, True, None)
    got = out.float().cpu().numpy().reshape(-1, D)[rows]
    assert_close(got, ref, *fwd_tol("bf16", v.float().abs().max().item()), "sampled rows")
    assert_close(lse.cpu().numpy().reshape(-1)[rows], ref_lse, LSE_TOL["bf16"], 1e-5, "LSE")
    # autograd round trip on a smaller instance of the same route
    S2 = 2048
    assert _route_causal("bf16", 1, 4, 4, S2, S2, D, True) == 7
    rng = np.random.RandomState(8)
    qn, kn, vn, don = (quantize(rng.randn(1, 4, S2, D).astype(np.float32), "bf16") for _ in range(4))
    tq, tk, tv = (torch.from_numpy(x).to("cuda", torch.bfloat16).requires_grad_(True) for x in (qn, kn, vn))
    o2 = aule.flash_attention(tq, tk, tv, causal=True)
    o2.backward(torch.from_numpy(don).to("cuda", torch.bfloat16))
    rq, rk, rv = oracle_mod.bwd_f64(qn, kn, vn, don, True)
    a, r = BWD_TOL["bf16"]
    for name, g, want in (("dq", tq.grad, rq), ("dk", tk.grad, rk), ("dv", tv.grad, rv)):
        assert_close(g.float().cpu().numpy(), want, a * max(1.0, float(np.abs(want).max())), r, name)
