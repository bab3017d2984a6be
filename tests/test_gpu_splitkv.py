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
