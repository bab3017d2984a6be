"""Bottom-right aligned causal mask on the GPU: `causal="bottom-right"` (C-ABI
AULE_CAUSAL_BOTTOM_RIGHT) puts query i at position i + Sk - Sq, so the last query sees every key.  An additive
option -- every reference implementation is top-left aligned, and `causal=True` keeps that rule.

  * forward and backward against the fp64 oracle (its `causal=2` mode) for 16-bit and fp32 I/O, with and without
    a sliding window, MHA/GQA/MQA, ragged sizes, short query chunks against a long key history;
  * properties: Sq == Sk is the top-left result bit for bit; a bottom-right chunk equals the matching rows of
    the square top-left problem (chunked prefill); Sk < Sq is rejected at both boundaries;
  * an independent torch fp32 masked reference.
"""
import ctypes
import math

import numpy as np
import pytest

from util import BWD_TOL, LSE_TOL, assert_close, fwd_tol, quantize, torch_dtype

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", torch_dtype(dtype))


CASES = [  # dtype, B, Hq, Hkv, Sq, Sk, D, window
    ("bf16", 1, 4, 2, 256, 640, 128, -1),
    ("bf16", 2, 2, 2, 1000, 1500, 128, -1),
    ("bf16", 1, 8, 1, 64, 2048, 128, -1),          # a short chunk against a long history (MQA)
    ("bf16", 1, 2, 2, 1, 777, 128, -1),            # one decode row: sees everything
    ("fp16", 1, 4, 4, 333, 500, 64, -1),
    ("fp16", 1, 3, 3, 130, 131, 32, -1),
    ("fp32", 1, 2, 1, 300, 420, 32, -1),
    ("fp32", 1, 2, 2, 77, 400, 128, -1),
    ("bf16", 1, 4, 2, 512, 1024, 128, 100),        # window measured from the shifted position
    ("bf16", 1, 4, 2, 512, 1024, 128, 200),         # round 6: windows of at least two key tiles run the window instances of the one-wave-per-SIMD kernel (route 8)
    ("bf16", 1, 2, 2, 300, 900, 64, 301),
    ("fp16", 1, 2, 1, 200, 1000, 128, 64),
    ("fp32", 1, 2, 2, 150, 400, 64, 33),
    ("bf16", 1, 2, 2, 2048, 2048 + 517, 128, -1),
    ("bf16", 2, 64, 16, 520, 1100, 128, -1),        # enough pairs of Q blocks to fill the chip: the plain tile stream
    # short chunks against a long history: the tiled kernel's SPLIT instances with the causal mask (route 5) --
    # packed rows whose waves span several heads, rows that see no key in the later key splits
    ("bf16", 2, 8, 2, 64, 8192, 128, -1),
    ("fp16", 1, 8, 2, 17, 3000, 64, -1),            # 68 packed rows
    ("fp16", 2, 6, 3, 33, 2049, 32, -1),            # 66 packed rows, ragged Sk
    ("bf16", 1, 32, 1, 5, 4100, 128, -1),           # MQA: 160 packed rows
    ("bf16", 8, 32, 8, 1, 4096, 128, -1),           # one query per sequence: routed as the non-causal problem
]

ROUTES = {  # forward kernel each of these shapes is meant to exercise (aule_hip_debug_forward_route)
    ("bf16", 2, 8, 2, 64, 8192, 128, -1): 5, ("fp16", 1, 8, 2, 17, 3000, 64, -1): 5, ("fp16", 2, 6, 3, 33, 2049, 32, -1): 5,
    ("bf16", 1, 32, 1, 5, 4100, 128, -1): 5, ("bf16", 8, 32, 8, 1, 4096, 128, -1): 4, ("bf16", 1, 8, 1, 64, 2048, 128, -1): 5,
    ("bf16", 2, 2, 2, 1000, 1500, 128, -1): 7, ("bf16", 2, 64, 16, 520, 1100, 128, -1): 8, ("bf16", 1, 4, 2, 512, 1024, 128, 100): 1, ("bf16", 1, 4, 2, 512, 1024, 128, 200): 8, ("fp32", 1, 2, 1, 300, 420, 32, -1): 0,
}


def _route(case):
    from aule import _capi
    dtype, B, Hq, Hkv, Sq, Sk, D, W = case
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = {"fp32": 0, "fp16": 1, "bf16": 2}[dtype]
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = 2, W
    return _capi.get_lib().aule_hip_debug_forward_route(ctypes.byref(d))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_bottom_right_forward_backward_vs_oracle(case, oracle_mod):
    import torch
    from aule import _torch as at
    dtype, B, Hq, Hkv, Sq, Sk, D, W = case
    if case in ROUTES:
        assert _route(case) == ROUTES[case], "the dispatch rule moved this shape off the kernel it was written for"
    rng = np.random.RandomState(23)
    q, k, v, do = (quantize(rng.randn(*s).astype(np.float32), dtype)
                   for s in ((B, Hq, Sq, D), (B, Hkv, Sk, D), (B, Hkv, Sk, D), (B, Hq, Sq, D)))
    sc = 1 / math.sqrt(D)
    tq, tk, tv, tdo = (_dev(torch, x, dtype) for x in (q, k, v, do))
    out, lse = at.fwd_raw(tq, tk, tv, "bottom-right", sc, window=W)
    ref, ref_lse = oracle_mod.fwd_f64(q, k, v, "bottom-right", None, W)
    atol, rtol = fwd_tol(dtype, np.abs(v).max())
    assert_close(out.float().cpu().numpy(), ref, atol, rtol, "out")
    assert np.all(np.isfinite(ref_lse))                 # every row sees at least its own position
    assert_close(lse.cpu().numpy(), ref_lse, LSE_TOL[dtype], 1e-5, "lse")
    dq, dk, dv = at.bwd_raw(tq, tk, tv, out, tdo, lse, "bottom-right", sc, window=W)
    rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, "bottom-right", None, W)
    a, r = BWD_TOL[dtype]
    for name, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert_close(got.float().cpu().numpy(), want, a * max(1.0, float(np.abs(want).max())), r, name)


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_bottom_right_vs_torch_masked_reference(dtype):
    """An independent check of the mask itself: plain torch fp32 softmax(QK^T + mask)V with autograd."""
    import torch
    import aule
    torch.manual_seed(5)
    td = torch_dtype(dtype)
    B, Hq, Hkv, Sq, Sk, D = 1, 4, 2, 200, 333, 64
    q = torch.randn(B, Hq, Sq, D, device="cuda", dtype=td, requires_grad=True)
    k = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=td, requires_grad=True)
    v = torch.randn(B, Hkv, Sk, D, device="cuda", dtype=td, requires_grad=True)
    do = torch.randn(B, Hq, Sq, D, device="cuda", dtype=td)
    out = aule.flash_attention(q, k, v, causal="bottom-right")
    out.backward(do)
    got = [out.detach().float(), q.grad.float(), k.grad.float(), v.grad.float()]
    qf, kf, vf = (x.detach().float().requires_grad_(True) for x in (q, k, v))
    i = torch.arange(Sq, device="cuda")[:, None] + (Sk - Sq)
    j = torch.arange(Sk, device="cuda")[None, :]
    s = (qf @ kf.repeat_interleave(Hq // Hkv, 1).transpose(-1, -2)) / math.sqrt(D)
    s = s.masked_fill(j > i, float("-inf"))
    ref = torch.softmax(s, -1) @ vf.repeat_interleave(Hq // Hkv, 1)
    ref.backward(do.float())
    tol = 3e-2 if dtype == "bf16" else 2e-4
    for name, g, w in zip(("out", "dq", "dk", "dv"), got, (ref.detach(), qf.grad, kf.grad, vf.grad)):
        err = float((g - w).abs().max())
        assert err < tol * max(1.0, float(w.abs().max())), (name, err)


def test_bottom_right_properties():
    import torch
    import aule
    torch.manual_seed(9)
    S, C, D = 1024, 192, 128
    q = torch.randn(1, 4, S, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 2, S, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(1, 2, S, D, device="cuda", dtype=torch.bfloat16)
    full = aule.flash_attention(q, k, v, causal=True)
    # Sq == Sk: both alignments are the same mask, and the same kernel path
    assert torch.equal(aule.flash_attention(q, k, v, causal="bottom-right"), full)
    # chunked prefill: the last C queries against all keys, bottom-right == those rows of the square problem
    tail = aule.flash_attention(q[:, :, S - C:].contiguous(), k, v, causal="bottom-right")
    assert torch.allclose(tail.float(), full[:, :, S - C:].float(), atol=1e-2, rtol=1e-2)
    # ... and a middle chunk against the keys up to its end
    lo, hi = 300, 300 + C
    mid = aule.flash_attention(q[:, :, lo:hi].contiguous(), k[:, :, :hi].contiguous(), v[:, :, :hi].contiguous(),
                               causal="bottom-right")
    assert torch.allclose(mid.float(), full[:, :, lo:hi].float(), atol=1e-2, rtol=1e-2)
    # causal="top-left" is causal=True
    assert torch.equal(aule.flash_attention(q, k, v, causal="top-left"), full)


def test_bottom_right_rejections():
    import torch
    import aule
    from aule import _capi
    q = torch.randn(1, 2, 64, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 2, 32, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        aule.flash_attention(q, k, k, causal="bottom-right")          # Sk < Sq
    with pytest.raises(ValueError):
        aule.flash_attention(q, k, k, causal="diagonal")
    lib = _capi.get_lib()
    out = torch.empty_like(q)
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = 2
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = 1, 2, 2, 64, 32, 64
    d.scale, d.window_size, d.device, d.stream = 0.125, -1, 0, None
    d.q, d.k, d.v, d.out, d.lse = q.data_ptr(), k.data_ptr(), k.data_ptr(), out.data_ptr(), None
    d.causal = 2
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3
    assert b"bottom-right" in lib.aule_get_error()
    d.causal = 3
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3
    assert b"causal mode" in lib.aule_get_error()
